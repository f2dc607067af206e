/**
 * TEST INFRASTRUCTURE — the drop-in boundary compiled against the REFERENCE'S OWN TYPES.
 *
 * INTEGRATION.md tells a Mara3 maintainer what to add to a sub-program. The code blocks it shows are cut out of THIS file (between the
 * `[integration:...]` markers; tests/test_integration_doc_cpu.py checks that the document and the file agree), and this file is
 * compiled against /root/reference/src (core_ndarray.hpp, physics_euler.hpp, physics_srhd.hpp, ...) and include/mara_hip.h and linked
 * with libmara_hip.so. So "drops in under subprog_*" is a statement about a program that builds and runs:
 *
 *   integration_ref euler <n0> <n1> <nsteps>     the lazy-array composition of advance / next_solution (euler_cart_compose.hpp: the
 *       reference's operators in the order of subprog_cloud.cpp:511-584, :676-697) on an nd::shared_array<conserved_density_t, 2>, and
 *       gpu_evaluator_t - the binding below - on THE SAME ARRAY OBJECT; the two results are compared bit for bit.
 *   integration_ref cloud <nr> <nsteps>          CloudProblem's state (cloud_compose.hpp: models, nd::linspace vertices, initial
 *       conserved array as new_solution builds them, subprog_cloud.cpp:610-662), stepped by the reference composition and by
 *       cloud_gpu_evaluator_t with the nozzle row evaluated by the reference's own jet model; compared bit for bit.
 *
 * exit code: 0 bit-identical, 1 different, 77 no GPU (the binding itself needs none to COMPILE AND LINK - that is the CPU test).
 */
#include "euler_cart_compose.hpp"
#include <stdexcept>

// [integration:euler begin]
#include "mara_hip.h"                       // extern "C"
#include "core_ndarray.hpp"
#include "physics_euler.hpp"

struct gpu_evaluator_t                      // plays the role of mara::evaluate_on<N>() for the whole step
{
    mh_ctx* ctx = nullptr;
    explicit gpu_evaluator_t(nd::shape_t<2> shape, double dx, double dy, double gamma, double plm_theta, int rk_order)
    {
        if (mh_create(&ctx, 0)) throw std::runtime_error(mh_last_error(nullptr));
        mh_euler_cart_desc d = {};
        d.rank = 2; d.n[0] = int(shape[0]); d.n[1] = int(shape[1]); d.n[2] = 1;
        d.dl[0] = dx; d.dl[1] = dy; d.dl[2] = 1.0;
        d.gamma = gamma; d.plm_theta = plm_theta; d.riemann = MH_RIEMANN_HLLE;
        d.bc_lo0 = d.bc_hi0 = d.bc_transverse = MH_BC_OUTFLOW;          // nd::extend_zero_gradient
        d.arith = MH_ARITH_STRICT;                                       // bit-identical to the CPU path
        if (mh_euler_cart_configure(ctx, &d, rk_order)) throw std::runtime_error(mh_last_error(ctx));
    }
    ~gpu_evaluator_t() { mh_destroy(ctx); }

    // solution.conserved is row-major AoS of double[5] (sizeof(conserved_density_t) == 40,
    // core_ndarray.hpp:777-792, physics_euler.hpp:46): hand its buffer over as it is.
    void upload(const nd::shared_array<mara::euler::conserved_density_t, 2>& u)
    {
        static_assert(sizeof(mara::euler::conserved_density_t) == 5 * sizeof(double), "cell layout");
        if (mh_upload(ctx, reinterpret_cast<const double*>(u.data()), u.size())) throw std::runtime_error(mh_last_error(ctx));
    }
    auto download(nd::shape_t<2> shape)
    {
        auto u = nd::make_unique_array<mara::euler::conserved_density_t>(shape);
        if (mh_download(ctx, reinterpret_cast<double*>(u.data()), u.size())) throw std::runtime_error(mh_last_error(ctx));
        return std::move(u).shared();
    }
    void step(double dt, int nsteps)         // replaces: s0*0.5 + advance(advance(s0))*0.5  (subprog_cloud.cpp:682-695)
    {
        if (mh_step(ctx, dt, nsteps)) throw std::runtime_error(mh_last_error(ctx));
        mh_step_result r;                    // device kernels cannot throw: {status bits, first failing flat cell index}
        mh_status(ctx, &r);
        if (r.status) throw std::runtime_error("unphysical state in cell " + std::to_string(r.first_bad_index));
    }
    // a caller that retries from the OLD solution (binary's safe mode, subprog_binary.cpp:285-292) uses the transactional form:
    bool try_step(double dt)                 // false: the step was rejected and the previous solution is still in place, bit for bit
    {
        mh_step_result r;
        const int rc = mh_step_checked(ctx, dt, &r);
        if (rc == MH_E_PHYSICS) return false;
        if (rc) throw std::runtime_error(mh_last_error(ctx));
        return true;
    }
};
// [integration:euler end]

#include "cloud_compose.hpp"

// [integration:cloud begin]
#include "physics_srhd.hpp"

struct cloud_gpu_evaluator_t                // CloudProblem::advance + next_solution (subprog_cloud.cpp:511-584, :676-697) on the device
{
    mh_ctx* ctx = nullptr;
    // radial_vertices: nd::shared_array<mara::unit_length<double>, 1>, polar_vertices: nd::shared_array<double, 1>, as solution_t holds
    // them (:101-108); a dimensional value is one double (core_dimensional.hpp:93-268), so both buffers go over as they are
    cloud_gpu_evaluator_t(const nd::shared_array<mara::unit_length<double>, 1>& radial_vertices, const nd::shared_array<double, 1>& polar_vertices,
                          double plm_theta, double temperature_floor, int rk_order)
    {
        static_assert(sizeof(mara::unit_length<double>) == sizeof(double), "a dimensional value is its number");
        if (mh_create(&ctx, 0)) throw std::runtime_error(mh_last_error(nullptr));
        mh_cloud_desc d = {};
        d.nr = d.nr_global = int(radial_vertices.size()) - 1;
        d.nq = int(polar_vertices.size()) - 1;
        d.row_offset = 0;
        d.gamma = 4. / 3;                                                 // #define gamma_law_index (:52)
        d.plm_theta = plm_theta;                                         // reconstruct_method 2; < 0 for method 1 (piecewise constant)
        d.temperature_floor = temperature_floor;
        d.bc_lo0 = MH_BC_INFLOW; d.bc_hi0 = MH_BC_OUTFLOW;               // nozzle row (:466-493) / zero gradient (:503-509)
        d.arith = MH_ARITH_STRICT;
        if (mh_cloud_configure(ctx, &d, reinterpret_cast<const double*>(radial_vertices.data()), polar_vertices.data(), rk_order))
            throw std::runtime_error(mh_last_error(ctx));
    }
    ~cloud_gpu_evaluator_t() { mh_destroy(ctx); }

    void upload(const nd::shared_array<mara::srhd::conserved_t, 2>& u)       // cell-integrated conserved, [nr][nq] of double[5]
    {
        static_assert(sizeof(mara::srhd::conserved_t) == 5 * sizeof(double), "cell layout");
        if (mh_upload(ctx, reinterpret_cast<const double*>(u.data()), u.size())) throw std::runtime_error(mh_last_error(ctx));
    }
    auto download(nd::shape_t<2> shape)
    {
        auto u = nd::make_unique_array<mara::srhd::conserved_t>(shape);
        if (mh_download(ctx, reinterpret_cast<double*>(u.data()), u.size())) throw std::runtime_error(mh_last_error(ctx));
        return std::move(u).shared();
    }
    // one time step; inflow_row = the primitives of the inner ghost cells at the step-START time, the array advance builds from the jet
    // model (:466-493) - evaluated by the caller with the reference's own model, as upstream, once per step for both RK stages (:524)
    void step(const nd::shared_array<mara::srhd::primitive_t, 1>& inflow_row, double dt)
    {
        static_assert(sizeof(mara::srhd::primitive_t) == 5 * sizeof(double), "primitive layout");
        if (mh_cloud_set_inflow(ctx, reinterpret_cast<const double*>(inflow_row.data()))) throw std::runtime_error(mh_last_error(ctx));
        if (mh_step(ctx, dt, 1)) throw std::runtime_error(mh_last_error(ctx));
        mh_step_result r;
        mh_status(ctx, &r);                  // one bit per `throw` of mara::srhd::recover_primitive (physics_srhd.hpp:430-449)
        if (r.status & MH_STATUS_NEG_DENSITY) throw std::invalid_argument("mara::srhd::recover_primitive failure: negative density");
        if (r.status) throw std::invalid_argument("mara::srhd::recover_primitive failure in cell " + std::to_string(r.first_bad_index));
    }
};
// [integration:cloud end]


//=============================================================================
template<typename A, typename B>
static int compare(const A& a, const B& b, const char* what)
{
    if (a.size() != b.size()) { std::printf("%s: sizes differ\n", what); return 1; }
    std::size_t bad = 0;
    if (std::memcmp(a.data(), b.data(), a.size() * 5 * sizeof(double)) != 0)
    {
        auto pa = reinterpret_cast<const double*>(a.data());
        auto pb = reinterpret_cast<const double*>(b.data());
        for (std::size_t i = 0; i < a.size() * 5; ++i) if (std::memcmp(pa + i, pb + i, sizeof(double)) != 0) ++bad;
    }
    std::printf("%s: %zu cells, %zu values differ from the reference composition%s\n", what, std::size_t(a.size()), bad, bad ? "" : " (bit-identical)");
    return bad ? 1 : 0;
}

static int run_euler(std::size_t n0, std::size_t n1, int nsteps)
{
    using namespace euler_cart;
    auto par = params_t();
    par.gamma = 5. / 3; par.theta = 1.5; par.bc = 0;
    par.dl[0] = 1.0 / n0; par.dl[1] = 1.0 / n1; par.dl[2] = 1.0;
    par.dt = 0.3 * std::min(par.dl[0], par.dl[1]) / 6;
    auto shape = nd::make_shape(n0, n1);

    // SURVEY.md section 8d's blast, as a lazy array of the reference's own types
    auto initial = nd::make_array([=] (auto ij)
    {
        auto x = (ij[0] + 0.5) / n0 - 0.5, y = (ij[1] + 0.5) / n1 - 0.5;
        auto p = prim_t().with_mass_density(1.0).with_gas_pressure(x * x + y * y < 0.01 ? 10.0 : 0.1);
        return p.to_conserved_density(par.gamma);
    }, shape);
    auto u0 = cons_array_t<2>(initial | nd::to_shared());

    if (mh_device_count() < 1) { std::printf("no GPU: the binding compiled and linked, nothing to run\n"); return 77; }

    auto s = u0;
    for (int n = 0; n < nsteps; ++n)
    {
        auto s2 = advance<2>(advance<2>(s, par), par);
        s = (s * 0.5 + s2 * 0.5) | nd::to_shared();
    }
    auto gpu = gpu_evaluator_t(shape, par.dl[0], par.dl[1], par.gamma, par.theta, 2);
    gpu.upload(u0);
    gpu.step(par.dt, nsteps);
    auto g = gpu.download(shape);
    int rc = compare(s, g, "euler: gpu_evaluator_t");
    // the transactional form on a healthy state commits and lands on the same bits
    auto gpu2 = gpu_evaluator_t(shape, par.dl[0], par.dl[1], par.gamma, par.theta, 2);
    gpu2.upload(u0);
    for (int n = 0; n < nsteps; ++n) if (! gpu2.try_step(par.dt)) { std::printf("try_step rejected a healthy step\n"); return 1; }
    return rc | compare(s, gpu2.download(shape), "euler: try_step");
}

static int run_cloud(int nr, int nsteps)
{
    using namespace cloud_compose;
    auto S = setup_t();
    double ref_length = 0.0, ref_mass = 0.0;
    auto u0 = make_cloud_problem(S, nr, 1.0, ref_length, ref_mass);
    auto dt = (S.rv | nd::difference_on_axis(0) | nd::read_index(0)) / mara::make_velocity(1.0) * 0.4;       // subprog_cloud.cpp:678-679
    auto shape = nd::make_shape(S.rv.size() - 1, S.qv.size() - 1);

    if (mh_device_count() < 1) { std::printf("no GPU: the binding compiled and linked, nothing to run\n"); return 77; }

    auto gpu = cloud_gpu_evaluator_t(S.rv, S.qv, S.theta, S.temperature_floor, 2);
    gpu.upload(u0);
    auto u = u0;
    double time = 0.0;
    for (int n = 0; n < nsteps; ++n)
    {
        auto inflow = std::vector<double>();
        auto s1 = advance(S, u, time, dt, &inflow);
        auto s2 = advance(S, s1, time, dt, nullptr);
        u = (u * 0.5 + s2 * (1 - 0.5)) | nd::to_shared();
        // the same nozzle row, as the array of primitives `advance` builds
        auto nq = shape[1];
        auto row = nd::make_array([&inflow] (auto j) { auto p = prim_t(); for (std::size_t q = 0; q < 5; ++q) p[q] = inflow[j[0] * 5 + q]; return p; }, nd::make_shape(nq)) | nd::to_shared();
        gpu.step(row, dt.value);
        time += dt.value;
    }
    return compare(u, gpu.download(shape), "cloud: cloud_gpu_evaluator_t");
}

int main(int argc, char** argv)
{
    try
    {
        if (argc == 5 && std::string(argv[1]) == "euler") return run_euler(std::atol(argv[2]), std::atol(argv[3]), std::atoi(argv[4]));
        if (argc == 4 && std::string(argv[1]) == "cloud") return run_cloud(std::atoi(argv[2]), std::atoi(argv[3]));
    }
    catch (const std::exception& e)
    {
        std::fprintf(stderr, "integration_ref: %s\n", e.what());
        return 2;
    }
    std::fprintf(stderr, "usage: integration_ref euler <n0> <n1> <nsteps> | cloud <nr> <nsteps>\n");
    return 2;
}
