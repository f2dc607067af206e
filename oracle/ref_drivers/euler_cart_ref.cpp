/**
 * TEST INFRASTRUCTURE — reference-side oracle driver (never shipped, never timed as product).
 *
 * Composes the *reference's own headers* (compiled where they lie, under
 * /root/reference/src) into a uniform-cartesian Euler PLM+HLLE step, the way
 * the reference composes its spherical SRHD step in
 * subprog_cloud.cpp:511-584 (advance) and :676-697 (next_solution, RK combine):
 *
 *     p0 = u0 | map(recover_primitive)                         (physics_euler.hpp:555-575)
 *     pe = p0 | extend_{zero_gradient,periodic}(axis)          (core_ndarray_ops.hpp:152-190)
 *     G  = pe | zip_adjacent3_on_axis | plm_gradient [| extend_zeros]   (math_interpolation.hpp:85-94)
 *     F  = zip(PL + GL*0.5, PR - GR*0.5) | riemann_hlle        (physics_euler.hpp:614-631)
 *     u1 = u0 - (diff(Fx)*(dt/dx) + diff(Fy)*(dt/dy) [+ diff(Fz)*(dt/dz)])
 *     RK2: s0*0.5 + advance(advance(s0))*0.5
 *
 * No such sub-program exists upstream; this driver is the
 * oracle for BASELINE configs 2 and 5. Only its OUTPUTS are committed (tests/golden).
 *
 * usage: euler_cart_ref <rank> <n0> <n1> <n2> <gamma> <theta> <rk> <bc> <dt> <d0> <d1> <d2> <nsteps> <in.f64> <out.f64> [threads]
 *        bc: 0 = zero-gradient (outflow), 1 = periodic; theta<0 => piecewise constant
 *        threads (PLM, bc 0): 1, 2, 4, 8, 16 or 32 = the step evaluated as upstream evaluates its own (lazy arrays, primitives and result
 *        through mara::evaluate_on<threads>(), src/app_parallel.hpp:72-103; subprog_cloud.cpp:525-533, :582); -1 = the same composition
 *        through nd::to_shared(). The bits do not depend on it (tests/test_oracle_golden.py): this is the reference's own CPU path for
 *        bench.py's cpu_reference.
 *        in/out: row-major AoS [n0][n1][n2][5] doubles (conserved densities)
 */
#include "euler_cart_compose.hpp"
using namespace euler_cart;

template<std::size_t Rank>
static int run(nd::shape_t<Rank> shape, params_t par, int rk, int nsteps, const char* fin, const char* fout, int threads)
{
    auto u = nd::make_unique_array<cons_t>(shape);
    auto ncell = shape.volume();
    static_assert(sizeof(cons_t) == 5 * sizeof(double), "cell layout");

    FILE* f = std::fopen(fin, "rb");
    if (! f || std::fread(u.data(), sizeof(cons_t), ncell, f) != ncell) { std::fprintf(stderr, "bad input %s\n", fin); return 2; }
    std::fclose(f);

    auto s = cons_array_t<Rank>(std::move(u).shared());

    if (threads != 0)
    {
        if (par.theta < 0.0 || par.bc != 0) { std::fprintf(stderr, "threads: PLM with zero-gradient sides only\n"); return 2; }
        s = with_upstream_evaluator(threads, [&] (auto evaluate)
        {
            auto t = s;
            for (int n = 0; n < nsteps; ++n)
            {
                if (rk == 1) t = advance_as_upstream_evaluates<Rank>(t, par, evaluate);
                else
                {
                    auto t2 = advance_as_upstream_evaluates<Rank>(advance_as_upstream_evaluates<Rank>(t, par, evaluate), par, evaluate);
                    t = (t * 0.5 + t2 * 0.5) | evaluate;          // next_solution, subprog_cloud.cpp:682-695
                }
            }
            return t;
        });
        nsteps = 0;
    }
    for (int n = 0; n < nsteps; ++n)
    {
        if (rk == 1)
        {
            s = advance<Rank>(s, par);
        }
        else
        {
            auto s2 = advance<Rank>(advance<Rank>(s, par), par);
            s = (s * 0.5 + s2 * 0.5) | nd::to_shared();
        }
    }
    FILE* g = std::fopen(fout, "wb");
    std::fwrite(s.data(), sizeof(cons_t), ncell, g);
    std::fclose(g);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc != 16 && argc != 17) { std::fprintf(stderr, "usage: see header\n"); return 1; }
    const int threads = argc == 17 ? std::atoi(argv[16]) : 0;
    int rank = std::atoi(argv[1]);
    std::size_t n0 = std::atol(argv[2]), n1 = std::atol(argv[3]), n2 = std::atol(argv[4]);
    params_t par;
    par.gamma = std::atof(argv[5]);
    par.theta = std::atof(argv[6]);
    int rk    = std::atoi(argv[7]);
    par.bc    = std::atoi(argv[8]);
    par.dt    = std::atof(argv[9]);
    par.dl[0] = std::atof(argv[10]);
    par.dl[1] = std::atof(argv[11]);
    par.dl[2] = std::atof(argv[12]);
    int nsteps = std::atoi(argv[13]);

    switch (rank)
    {
        case 1: return run<1>(nd::make_shape(n0), par, rk, nsteps, argv[14], argv[15], threads);
        case 2: return run<2>(nd::make_shape(n0, n1), par, rk, nsteps, argv[14], argv[15], threads);
        case 3: return run<3>(nd::make_shape(n0, n1, n2), par, rk, nsteps, argv[14], argv[15], threads);
    }
    return 1;
}
