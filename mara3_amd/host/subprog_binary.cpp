// `binary` sub-program: isothermal circumbinary disk (BASELINE config 3), the compiled host of the reference's
// src/subprog_binary.cpp over the C ABI. Same run_config items and defaults (create_config_template :55-99), same run
// loop (`while simulation_should_continue: next_state`, :414-438, with time measured per iteration and the
// `[iter] orbits=... kzps=...` message of :394-404), same set-up (vertices :165-185, disk model :105-153, solver data
// subprog_binary_solver_data.cpp:20-102) - evaluated by the library's host functions with the host libm - and the same
// step semantics including the safe-mode retry (:258-293), which the library performs on the device-resident state.
//
// The tree is built as upstream (refinement predicate + 2:1 balance, mh_binary_tree_build). A tree of uniform depth runs as
// one periodic grid through the wave-marching kernels (binary.hip); a graded one - the default, focus_factor=2 - block by
// block with prolonged / restricted guard zones and flux correction (binary_tree.hip).
// Restrictions, stated rather than silently ignored: the HDF5 tasks (checkpoint, diagnostics, time series: cpi, dfi,
// tsi) are out of scope for `binary`; a raw dump of the final state replaces them.
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
    .item("restart",             "")
    .item("outdir",              "data")
    .item("cpi",                 10.0)
    .item("dfi",                  1.0)
    .item("tsi",                 2e-3)
    .item("tfinal",               1.0)
    .item("cfl_number",           0.4)
    .item("fixed_dt",               0)
    .item("depth",                  4)
    .item("begin_live_binary",    1e6)
    .item("conserve_linear_p",      1)
    .item("block_size",            24)
    .item("focus_factor",        2.00)
    .item("focus_index",         2.00)
    .item("threaded",               1)          // accepted for command-line compatibility; the device replaces the thread pool
    .item("rk_order",               2)
    .item("reconstruct_method", "plm")
    .item("plm_theta",            1.8)
    .item("source_term_softening", 1.)
    .item("softening_radius",    0.05)
    .item("sink_radius",         0.05)
    .item("sink_rate",            1.0)
    .item("buffer_damping_rate", 10.0)
    .item("domain_radius",       12.0)
    .item("disk_radius",          2.0)
    .item("disk_mass",           1e-3)
    .item("ambient_density",     1e-4)
    .item("density_floor",        0.0)
    .item("separation",           1.0)
    .item("mass_ratio",           1.0)
    .item("eccentricity",         0.0)
    .item("counter_rotate",         0)
    .item("mach_number",         10.0)
    .item("axisymmetric_cs2",       0)
    .item("no_accretion_force",     0)
    .item("alpha_cutoff_radius",  0.0)
    .item("alpha",                0.1)
    .item("nu",                   0.0)
    .item("mdot",                 0.0)
    // not upstream:
    .item("arith",           "strict")          // strict | fast (see include/mara_hip.h)
    .item("max_iterations",         0)          // stop after this many iterations (0 = run to tfinal)
    .item("steps_per_call",         1)          // iterations per mh_binary_next call (the state stays on the device either way)
    .item("write_final",            1)
    .item("device",                 0);
}

void check(int rc, const char* what)
{
    if (rc != MH_OK) throw std::runtime_error(std::string(what) + ": " + mh_last_error(nullptr));
}

class subprog_binary : public mara::sub_program_t
{
public:
    int main(int argc, const char* argv[]) override
    {
        auto cfg = config_template().update(argc, argv);
        cfg.pretty_print(stdout, "config");
        if (! cfg.get_string("restart").empty()) throw std::invalid_argument("binary: restart from an HDF5 checkpoint is out of scope in this build");
        if (cfg.get_string("reconstruct_method") != "plm" && cfg.get_string("reconstruct_method") != "pcm")
            throw std::invalid_argument("invalid reconstruct_method '" + cfg.get_string("reconstruct_method") + "', must be plm or pcm");
        const int depth = cfg.get_int("depth"), bs = cfg.get_int("block_size");
        const int nblocks = mh_binary_tree_build(bs, depth, cfg.get_double("focus_factor"), cfg.get_double("focus_index"), nullptr, 0);
        if (nblocks < 0) throw std::invalid_argument("binary: cannot build the block tree (block_size must be even, depth <= 12)");
        std::vector<mh_tree_block> blocks(nblocks);
        mh_binary_tree_build(bs, depth, cfg.get_double("focus_factor"), cfg.get_double("focus_index"), blocks.data(), nblocks);
        bool graded = false;
        for (const auto& b : blocks) graded = graded || b.level != blocks[0].level;
        if (graded) return run_graded(cfg, blocks);
        const int n = bs << blocks[0].level;
        if (blocks[0].level != depth) throw std::invalid_argument("binary: the refinement predicate stops the tree above the requested depth");

        std::vector<double> xv(n + 1), u(std::size_t(3) * n * n), br(std::size_t(n) * n);
        check(mh_binary_vertices(bs, depth, cfg.get_double("domain_radius"), xv.data()), "mh_binary_vertices");
        mh_binary_model model = {};
        model.softening_radius = cfg.get_double("softening_radius");
        model.disk_radius = cfg.get_double("disk_radius");
        model.mach_number = cfg.get_double("mach_number");
        model.disk_mass = cfg.get_double("disk_mass");
        model.ambient_density = cfg.get_double("ambient_density");
        model.mdot = cfg.get_double("mdot");
        model.counter_rotate = cfg.get_int("counter_rotate");
        model.angmom_form = cfg.get_int("conserve_linear_p") ? 0 : 1;
        model.buffer_damping_rate = cfg.get_double("buffer_damping_rate");
        model.domain_radius = cfg.get_double("domain_radius");
        model.cfl_number = cfg.get_double("cfl_number");
        mh_binary_run run = {};
        check(mh_binary_solver_data(&model, n, xv.data(), xv.data(), u.data(), br.data(), &run.recommended_time_step), "mh_binary_solver_data");
        run.rk_order = cfg.get_int("rk_order");
        run.fixed_dt = cfg.get_int("fixed_dt");
        run.no_accretion_force = cfg.get_int("no_accretion_force");
        run.cfl_number = cfg.get_double("cfl_number");
        run.begin_live_binary = cfg.get_double("begin_live_binary");

        mh_binary_desc d = {};
        d.n = n;
        d.block_size = bs;
        d.domain_radius = cfg.get_double("domain_radius");
        d.mach_number = cfg.get_double("mach_number");
        d.alpha = cfg.get_double("alpha");
        d.nu = cfg.get_double("nu");
        d.alpha_cutoff_radius = cfg.get_double("alpha_cutoff_radius");
        d.sink_rate = cfg.get_double("sink_rate");
        d.sink_radius = cfg.get_double("sink_radius");
        d.softening_radius = cfg.get_double("softening_radius");
        d.density_floor = cfg.get_double("density_floor") * cfg.get_double("disk_mass");
        d.axisymmetric_cs2 = cfg.get_int("axisymmetric_cs2");
        d.angmom_form = model.angmom_form;
        d.arith = cfg.get_string("arith") == "fast" ? MH_ARITH_FAST : MH_ARITH_STRICT;
        {
            double min_d = xv[1] - xv[0];                                  // solver_data.cpp:37-52, :91 (both axes share the vertex array)
            for (int i = 0; i < n; ++i) min_d = std::min(min_d, xv[i + 1] - xv[i]);
            d.gst_suppr_radius = cfg.get_double("source_term_softening") * min_d;
        }
        d.plm_theta = cfg.get_double("plm_theta");       // upstream validates reconstruct_method (solver_data.cpp:110-112) but the scheme never reads it

        mh_binary* solver = nullptr;
        check(mh_binary_create(&solver, cfg.get_int("device"), &d, &run, xv.data(), xv.data(), u.data(), br.data()), "mh_binary_create");
        mh_binary_state state = {};      // binary::create_solution :197-229
        state.orbital_elements.elements.total_mass = 1.0;
        state.orbital_elements.elements.separation = cfg.get_double("separation");
        state.orbital_elements.elements.mass_ratio = cfg.get_double("mass_ratio");
        state.orbital_elements.elements.eccentricity = cfg.get_double("eccentricity");
        check(mh_binary_set_solution(solver, nullptr, &state), "mh_binary_set_solution");

        const double tfinal = cfg.get_double("tfinal");
        const int batch = std::max(1, cfg.get_int("steps_per_call"));
        const int max_iter = cfg.get_int("max_iterations");
        while (state.time / (2 * M_PI) < tfinal && (max_iter == 0 || state.iteration < max_iter))
        {
            const int todo = max_iter ? int(std::min<long>(batch, max_iter - state.iteration)) : batch;
            int safe = 0;
            const double ms = host::time_ms([&] { check(mh_binary_next(solver, todo, &safe), "mh_binary_next"); });
            check(mh_binary_get_solution(solver, nullptr, &state), "mh_binary_get_solution");
            if (safe) std::printf("negative density in updated state\n");        // what the reference prints before its safe-mode retry
            std::printf("[%04ld] orbits=%3.7lf kzps=%3.2lf\n", long(state.iteration), state.time / (2 * M_PI), double(n) * n * todo / ms);
            std::fflush(stdout);
        }
        if (cfg.get_int("write_final"))
        {
            check(mh_binary_get_solution(solver, u.data(), &state), "mh_binary_get_solution");
            const std::vector<double> scalars = {
                state.mass_accreted_on[0], state.mass_accreted_on[1], state.angular_momentum_accreted_on[0], state.angular_momentum_accreted_on[1],
                state.integrated_torque_on[0], state.integrated_torque_on[1], state.work_done_on[0], state.work_done_on[1],
                state.mass_ejected, state.angular_momentum_ejected};
            std::vector<double> extra = xv;
            extra.insert(extra.end(), scalars.begin(), scalars.end());
            host::dump_state(cfg.get_string("outdir"), "final.bin", {long(n), long(n)}, 3, state.time, state.iteration, extra, u);
        }
        mh_binary_destroy(solver);
        return 0;
    }

    // the same run loop on a graded tree (block-major arrays)
    int run_graded(const mara::config_t& cfg, const std::vector<mh_tree_block>& blocks)
    {
        const int bs = cfg.get_int("block_size"), nb = int(blocks.size());
        std::vector<double> edges(std::size_t(nb) * 2 * (bs + 1)), u(std::size_t(nb) * bs * bs * 3), br(std::size_t(nb) * bs * bs);
        check(mh_binary_tree_vertices(bs, cfg.get_double("domain_radius"), blocks.data(), nb, edges.data()), "mh_binary_tree_vertices");
        mh_binary_model model = {};
        model.softening_radius = cfg.get_double("softening_radius");
        model.disk_radius = cfg.get_double("disk_radius");
        model.mach_number = cfg.get_double("mach_number");
        model.disk_mass = cfg.get_double("disk_mass");
        model.ambient_density = cfg.get_double("ambient_density");
        model.mdot = cfg.get_double("mdot");
        model.counter_rotate = cfg.get_int("counter_rotate");
        model.angmom_form = cfg.get_int("conserve_linear_p") ? 0 : 1;
        model.buffer_damping_rate = cfg.get_double("buffer_damping_rate");
        model.domain_radius = cfg.get_double("domain_radius");
        model.cfl_number = cfg.get_double("cfl_number");
        mh_binary_run run = {};
        check(mh_binary_tree_solver_data(&model, bs, blocks.data(), nb, edges.data(), u.data(), br.data(), &run.recommended_time_step), "mh_binary_tree_solver_data");
        run.rk_order = cfg.get_int("rk_order");
        run.fixed_dt = cfg.get_int("fixed_dt");
        run.no_accretion_force = cfg.get_int("no_accretion_force");
        run.cfl_number = cfg.get_double("cfl_number");
        run.begin_live_binary = cfg.get_double("begin_live_binary");
        mh_binary_desc d = {};
        d.block_size = bs;
        d.domain_radius = cfg.get_double("domain_radius");
        d.mach_number = cfg.get_double("mach_number");
        d.alpha = cfg.get_double("alpha");
        d.nu = cfg.get_double("nu");
        d.alpha_cutoff_radius = cfg.get_double("alpha_cutoff_radius");
        d.sink_rate = cfg.get_double("sink_rate");
        d.sink_radius = cfg.get_double("sink_radius");
        d.softening_radius = cfg.get_double("softening_radius");
        d.density_floor = cfg.get_double("density_floor") * cfg.get_double("disk_mass");
        d.axisymmetric_cs2 = cfg.get_int("axisymmetric_cs2");
        d.arith = cfg.get_string("arith") == "fast" ? MH_ARITH_FAST : MH_ARITH_STRICT;
        d.plm_theta = cfg.get_double("plm_theta");
        d.angmom_form = model.angmom_form;
        {
            double min_d = 1e300;                          // smallest vertex spacing of any block: solver_data.cpp:37-52, :91
            for (int k = 0; k < nb; ++k)
                for (int ax = 0; ax < 2; ++ax)
                    for (int a = 0; a < bs; ++a)
                        min_d = std::min(min_d, edges[(std::size_t(k) * 2 + ax) * (bs + 1) + a + 1] - edges[(std::size_t(k) * 2 + ax) * (bs + 1) + a]);
            d.gst_suppr_radius = cfg.get_double("source_term_softening") * min_d;
        }

        mh_binary* solver = nullptr;
        check(mh_binary_tree_create(&solver, cfg.get_int("device"), &d, &run, blocks.data(), nb, edges.data(), u.data(), br.data()), "mh_binary_tree_create");
        mh_binary_state state = {};
        state.orbital_elements.elements.total_mass = 1.0;
        state.orbital_elements.elements.separation = cfg.get_double("separation");
        state.orbital_elements.elements.mass_ratio = cfg.get_double("mass_ratio");
        state.orbital_elements.elements.eccentricity = cfg.get_double("eccentricity");
        check(mh_binary_set_solution(solver, nullptr, &state), "mh_binary_set_solution");

        const double tfinal = cfg.get_double("tfinal"), cells = double(nb) * bs * bs;
        const int batch = std::max(1, cfg.get_int("steps_per_call"));
        const int max_iter = cfg.get_int("max_iterations");
        std::printf("block tree: %d blocks of %d x %d zones\n", nb, bs, bs);
        while (state.time / (2 * M_PI) < tfinal && (max_iter == 0 || state.iteration < max_iter))
        {
            const int todo = max_iter ? int(std::min<long>(batch, max_iter - state.iteration)) : batch;
            int safe = 0;
            const double ms = host::time_ms([&] { check(mh_binary_next(solver, todo, &safe), "mh_binary_next"); });
            check(mh_binary_get_solution(solver, nullptr, &state), "mh_binary_get_solution");
            if (safe) std::printf("negative density in updated state\n");
            std::printf("[%04ld] orbits=%3.7lf kzps=%3.2lf\n", long(state.iteration), state.time / (2 * M_PI), cells * todo / ms);
            std::fflush(stdout);
        }
        if (cfg.get_int("write_final"))
        {
            check(mh_binary_get_solution(solver, u.data(), &state), "mh_binary_get_solution");
            std::vector<double> extra;                 // block list (level, i, j as doubles), then the ten accumulators
            for (const auto& b : blocks) { extra.push_back(b.level); extra.push_back(b.i); extra.push_back(b.j); }
            for (double v : {state.mass_accreted_on[0], state.mass_accreted_on[1], state.angular_momentum_accreted_on[0], state.angular_momentum_accreted_on[1],
                             state.integrated_torque_on[0], state.integrated_torque_on[1], state.work_done_on[0], state.work_done_on[1],
                             state.mass_ejected, state.angular_momentum_ejected}) extra.push_back(v);
            host::dump_state(cfg.get_string("outdir"), "final.bin", {long(nb), long(bs), long(bs)}, 3, state.time, state.iteration, extra, u);
        }
        mh_binary_destroy(solver);
        return 0;
    }

    std::string name() const override { return "binary"; }
};

} // namespace

std::unique_ptr<mara::sub_program_t> make_subprog_binary() { return std::make_unique<subprog_binary>(); }
