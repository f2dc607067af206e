// `euler2d` sub-program: the uniform-cartesian 2-D Euler blast of BASELINE config 2
// (no such sub-program exists upstream; the scheme is cloud::advance's composition,
// src/subprog_cloud.cpp:511-584, with mara::euler traits on a cartesian grid, see
// DESIGN.md §2). Run loop, fixed time step and the per-iteration `kzps` message follow
// the reference drivers (src/subprog_cloud.cpp:676-697, :856-861, :929-933).
#include <cmath>
#include <cstdio>
#include <vector>
#include "app_config.hpp"
#include "app_subprogram.hpp"
#include "host_common.hpp"

namespace {

mara::config_t config_template()
{
    return mara::config_t()
    .item("outdir", "data")
    .item("n", 1024)                  // cells per axis
    .item("tfinal", 0.01)
    .item("gamma", 5. / 3)
    .item("cfl_number", 0.3)          // dt = cfl * dx / 6   
    .item("rk_order", 2)
    .item("reconstruct_method", 2)    // 1: piecewise constant, 2: PLM   (as in cloud, src/subprog_cloud.cpp:84)
    .item("plm_theta", 1.5)
    .item("riemann", "hlle")          // hlle | hllc
    .item("arith", "strict")          // strict | fast
    .item("blast_radius", 0.1)
    .item("blast_pressure", 10.0)
    .item("ambient_pressure", 0.1)
    .item("write_final", 1)
    .item("steps_per_call", 10)       // time steps per mh_step call (the reference steps one at a time; state stays on the device either way)
    .item("device", 0)
    .item("gpus", 1);                 // axis-0 slabs over this many GPUs of the node, one process (not a reference option)
}

class subprog_euler2d : public mara::sub_program_t
{
public:
    int main(int argc, const char* argv[]) override
    {
        auto cfg = config_template().update(argc, argv);
        cfg.pretty_print(stdout, "config");
        const int n = cfg.get_int("n");
        const double gamma = cfg.get_double("gamma");
        const double dx = 1.0 / n;

        std::vector<double> u(std::size_t(5) * n * n);
        const double r0 = cfg.get_double("blast_radius");
        for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
        {
            const double x = (i + 0.5) / n, y = (j + 0.5) / n;
            const double r2 = (x - 0.5) * (x - 0.5) + (y - 0.5) * (y - 0.5);
            const double p = r2 < r0 * r0 ? cfg.get_double("blast_pressure") : cfg.get_double("ambient_pressure");
            double* c = &u[(std::size_t(i) * n + j) * 5];
            c[0] = 1.0; c[1] = 0.0; c[2] = 0.0; c[3] = 0.0; c[4] = p / (gamma - 1.0);
        }

        mh_euler_cart_desc d = {};
        d.rank = 2;
        d.n[0] = n; d.n[1] = n; d.n[2] = 1;
        d.dl[0] = dx; d.dl[1] = dx; d.dl[2] = 1.0;
        d.gamma = gamma;
        d.plm_theta = cfg.get_int("reconstruct_method") == 1 ? -1.0 : cfg.get_double("plm_theta");
        d.riemann = cfg.get_string("riemann") == "hllc" ? MH_RIEMANN_HLLC : MH_RIEMANN_HLLE;
        d.bc_lo0 = d.bc_hi0 = d.bc_transverse = MH_BC_OUTFLOW;
        d.arith = cfg.get_string("arith") == "fast" ? MH_ARITH_FAST : MH_ARITH_STRICT;

        // gpus > 1: axis-0 slabs (nd::partition_shape, the cut of the reference's evaluate_on<N> thread slabs) on a group of devices driven
        // by this one process, two-row halo per stage; more slabs than visible devices share them round-robin
        const int gpus = int(cfg.get_int("gpus"));
        if (gpus < 1 || gpus > 64) throw std::invalid_argument("gpus must be 1..64");
        auto group_check = [] (int rc, const char* what) { if (rc != MH_OK) throw std::runtime_error(std::string(what) + ": " + mh_last_error(nullptr)); };
        mh_ctx* ctx = nullptr;
        std::vector<mh_slab*> slabs(gpus > 1 ? gpus : 0, nullptr);
        if (gpus > 1)
        {
            const int visible = mh_device_count();
            std::vector<int> ids(gpus);
            for (int r = 0; r < gpus; ++r) ids[r] = r % (visible > 0 ? visible : 1);
            if (visible < gpus) std::printf("gpus=%d on %d visible device(s): slabs share devices round-robin\n", gpus, visible);
            group_check(mh_slab_group_create_on(slabs.data(), &d, int(cfg.get_int("rk_order")), gpus, ids.data()), "mh_slab_group_create_on");
            group_check(mh_slab_group_upload(slabs.data(), gpus, u.data()), "mh_slab_group_upload");
        }
        else
        {
            host::check(mh_create(&ctx, cfg.get_int("device")), nullptr, "mh_create");
            host::check(mh_euler_cart_configure(ctx, &d, cfg.get_int("rk_order")), ctx, "mh_euler_cart_configure");
            host::check(mh_upload(ctx, u.data(), std::size_t(n) * n), ctx, "mh_upload");
        }

        const double dt = cfg.get_double("cfl_number") * dx / 6.0;
        const double tfinal = cfg.get_double("tfinal");
        const int batch = cfg.get_int("steps_per_call");
        double time = 0.0;
        long iteration = 0;

        while (time < tfinal)
        {
            int todo = 0;
            for (double t = time; t < tfinal && todo < batch; t += dt) ++todo;
            const double ms = host::time_ms([&] {
                if (gpus > 1)
                {
                    group_check(mh_slab_group_step(slabs.data(), gpus, dt, todo), "mh_slab_group_step");
                    for (mh_slab* sl : slabs) group_check(mh_slab_synchronize(sl), "mh_slab_synchronize");
                }
                else
                {
                    host::check(mh_step(ctx, dt, todo), ctx, "mh_step");
                    host::check(mh_synchronize(ctx), ctx, "mh_synchronize");
                }
            });
            for (int s = 0; s < todo; ++s) time += dt;
            iteration += todo;
            if (gpus > 1)
            {
                mh_step_result worst = {0, 0, UINT64_MAX};
                for (mh_slab* sl : slabs)
                {
                    mh_step_result r;
                    group_check(mh_slab_status(sl, &r), "mh_slab_status");
                    worst.status |= r.status;
                    if (r.status && r.first_bad_index < worst.first_bad_index) worst.first_bad_index = r.first_bad_index;
                }
                host::throw_on_result(worst);
            }
            else host::throw_on_status(ctx);
            std::printf("[%04ld] t=%3.7lf kzps=%3.2lf\n", iteration, time, double(n) * n * todo / ms);
        }
        if (cfg.get_int("write_final"))
        {
            if (gpus > 1) group_check(mh_slab_group_download(slabs.data(), gpus, u.data()), "mh_slab_group_download");
            else          host::check(mh_download(ctx, u.data(), std::size_t(n) * n), ctx, "mh_download");
            host::dump_state(cfg.get_string("outdir"), "final.bin", {long(n), long(n)}, 5, time, iteration, {}, u);
        }
        for (mh_slab* sl : slabs) mh_slab_destroy(sl);
        mh_destroy(ctx);
        return 0;
    }

    std::string name() const override { return "euler2d"; }
};

} // namespace

std::unique_ptr<mara::sub_program_t> make_subprog_euler2d() { return std::make_unique<subprog_euler2d>(); }
