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
 * usage: euler_cart_ref <rank> <n0> <n1> <n2> <gamma> <theta> <rk> <bc> <dt> <d0> <d1> <d2> <nsteps> <in.f64> <out.f64>
 *        bc: 0 = zero-gradient (outflow), 1 = periodic; theta<0 => piecewise constant
 *        in/out: row-major AoS [n0][n1][n2][5] doubles (conserved densities)
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <numeric>
#include <vector>
#include <string>
#include "core_ndarray.hpp"
#include "core_ndarray_ops.hpp"
#include "core_dimensional.hpp"
#include "core_sequence.hpp"
#include "math_interpolation.hpp"
#include "physics_euler.hpp"

using cons_t = mara::euler::conserved_density_t;
using prim_t = mara::euler::primitive_t;

struct params_t
{
    double gamma, theta, dt;
    double dl[3];
    int bc;
};

template<std::size_t Rank>
using cons_array_t = nd::shared_array<cons_t, Rank>;

template<std::size_t Rank>
static cons_array_t<Rank> advance(cons_array_t<Rank> u0, params_t par)
{
    auto c2p = [g=par.gamma] (cons_t U) { return mara::euler::recover_primitive(U, g, 0.0); };
    auto p0 = u0 | nd::map(c2p) | nd::to_shared();

    auto godunov_flux_difference = [&] (std::size_t axis)
    {
        auto nh = mara::unit_vector_t::on_axis(axis);
        auto riemann = [nh, g=par.gamma] (prim_t pl, prim_t pr) { return mara::euler::riemann_hlle(pl, pr, nh, g); };
        auto L = nd::select_axis(axis).from(0).to(1).from_the_end();
        auto R = nd::select_axis(axis).from(1).to(0).from_the_end();
        auto dtdl = mara::make_time(par.dt) / mara::make_length(par.dl[axis]);

        if (par.theta < 0.0) // piecewise constant
        {
            auto F = (par.bc == 1
                ? (p0 | nd::extend_periodic_on_axis(axis, 1) | nd::to_shared())
                : (p0 | nd::extend_zero_gradient(axis)       | nd::to_shared()))
            | nd::zip_adjacent2_on_axis(axis) | nd::apply(riemann) | nd::to_shared();
            return (F | nd::difference_on_axis(axis)) * dtdl | nd::to_shared();
        }

        auto plm = [t=par.theta] (prim_t a, prim_t b, prim_t c) { return mara::plm_gradient(a, b, c, t); };

        if (par.bc == 1)
        {
            auto pe = p0 | nd::extend_periodic_on_axis(axis, 2) | nd::to_shared();
            auto G  = pe | nd::zip_adjacent3_on_axis(axis) | nd::apply(plm) | nd::to_shared();
            auto pi = pe | nd::select_axis(axis).from(1).to(1).from_the_end() | nd::to_shared();
            auto F  = nd::zip((pi | L) + (G | L) * 0.5, (pi | R) - (G | R) * 0.5) | nd::apply(riemann) | nd::to_shared();
            return (F | nd::difference_on_axis(axis)) * dtdl | nd::to_shared();
        }
        auto pe = p0 | nd::extend_zero_gradient(axis) | nd::to_shared();
        auto G  = pe | nd::zip_adjacent3_on_axis(axis) | nd::apply(plm) | nd::extend_zeros(axis) | nd::to_shared();
        auto F  = nd::zip((pe | L) + (G | L) * 0.5, (pe | R) - (G | R) * 0.5) | nd::apply(riemann) | nd::to_shared();
        return (F | nd::difference_on_axis(axis)) * dtdl | nd::to_shared();
    };

    if constexpr (Rank == 1)
    {
        return (u0 - godunov_flux_difference(0)) | nd::to_shared();
    }
    else if constexpr (Rank == 2)
    {
        return (u0 - (godunov_flux_difference(0) + godunov_flux_difference(1))) | nd::to_shared();
    }
    else
    {
        return (u0 - (godunov_flux_difference(0) + godunov_flux_difference(1) + godunov_flux_difference(2))) | nd::to_shared();
    }
}

template<std::size_t Rank>
static int run(nd::shape_t<Rank> shape, params_t par, int rk, int nsteps, const char* fin, const char* fout)
{
    auto u = nd::make_unique_array<cons_t>(shape);
    auto ncell = shape.volume();
    static_assert(sizeof(cons_t) == 5 * sizeof(double), "cell layout");

    FILE* f = std::fopen(fin, "rb");
    if (! f || std::fread(u.data(), sizeof(cons_t), ncell, f) != ncell) { std::fprintf(stderr, "bad input %s\n", fin); return 2; }
    std::fclose(f);

    auto s = cons_array_t<Rank>(std::move(u).shared());

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
    if (argc != 16) { std::fprintf(stderr, "usage: see header\n"); return 1; }
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
        case 1: return run<1>(nd::make_shape(n0), par, rk, nsteps, argv[14], argv[15]);
        case 2: return run<2>(nd::make_shape(n0, n1), par, rk, nsteps, argv[14], argv[15]);
        case 3: return run<3>(nd::make_shape(n0, n1, n2), par, rk, nsteps, argv[14], argv[15]);
    }
    return 1;
}
