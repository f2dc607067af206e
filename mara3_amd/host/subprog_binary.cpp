// `binary` sub-program: isothermal circumbinary disk (BASELINE config 3), the compiled host of the reference's
// src/subprog_binary.cpp over the C ABI. Same run_config items and defaults (create_config_template :55-99), same run
// loop (`state = tasks(state); while simulation_should_continue: state = tasks(next(state))`, then one closing
// `tasks(next(state))`, :414-438, with time measured per iteration and the `[iter] orbits=... kzps=...` message of :394-404),
// same set-up (vertices :165-185, disk model :105-153, solver data subprog_binary_solver_data.cpp:20-102) - evaluated by the
// library's host functions with the host libm - and the same step semantics including the safe-mode retry (:258-293), which the
// library performs on the device-resident state.
//
// The tree is built as upstream (refinement predicate + 2:1 balance, mh_binary_tree_build). A tree of uniform depth runs as
// one periodic grid through the wave-marching kernels (binary.hip); a graded one - the default, focus_factor=2 - block by
// block with prolonged / restricted guard zones and flux correction (binary_tree.hip).
//
// Tasks (next_schedule :296-302, run_tasks :325-385), SURVEY.md §8 rows f-1 and f-4: write_diagnostics (diagnostics.NNNN.h5:
// sigma, radial and azimuthal velocity per block, computed on the device), record_time_series (a sample of the accumulators plus
// disk mass and angular momentum, reduced on the device) and write_checkpoint (chkpt.NNNN.h5: solution with one dataset per tree
// block named level:ii-jj, schedule, time series, run_config; subprog_binary_io.cpp:129-160, app_serialize_tree.hpp:74-90) at
// intervals of cpi / dfi / tsi orbits, and restart=<checkpoint>. The field only leaves the device when a file is written.
// File-format parity is UNPINNED (h5_checkpoint.hpp); one deliberate statement about it: the reference stores iso2d conserved
// states as the raw bytes of an arithmetic_tuple_t, i.e. in std::tuple's storage order, which libstdc++ reverses.
// Files are written the way a g++/libstdc++ build of the reference writes and reads them: [p_y, p_x, Sigma] (or [L_z, S_r, Sigma]).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "app_config.hpp"
#include "app_subprogram.hpp"
#include "h5_checkpoint.hpp"
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
    .item("max_iterations",         0)          // stop after this many iterations (0 = run to tfinal, then upstream's closing step for its tasks)
    .item("steps_per_call",         1)          // iterations per mh_binary_next call; used only when all three tasks are switched off (interval <= 0)
    .item("write_final",            1)
    .item("device",                 0);
}

void check(int rc, const char* what)
{
    if (rc != MH_OK) throw std::runtime_error(std::string(what) + ": " + mh_last_error(nullptr));
}

// binary::time_series_sample_t (subprog_binary.hpp:160-178), member for member: the compound type of /time_series takes its offsets from here
struct time_series_sample_t
{
    double time = 0.0;
    double disk_mass = 0.0;
    double disk_angular_momentum = 0.0;
    double mass_ejected = 0.0;
    double angular_momentum_ejected = 0.0;
    double mass_accreted_on[2] = {0, 0};
    double angular_momentum_accreted_on[2] = {0, 0};
    double integrated_torque_on[2] = {0, 0};
    double work_done_on[2] = {0, 0};
    mh_full_orbital_elements orbital_elements_acc = {};
    mh_full_orbital_elements orbital_elements_grav = {};
    mh_full_orbital_elements orbital_elements = {};
    double position_of_mass1[2] = {0, 0};
    double position_of_mass2[2] = {0, 0};
};
static_assert(sizeof(time_series_sample_t) == 47 * sizeof(double), "time_series_sample_t must be 47 packed doubles");

// hdf5_type_info<orbital_elements_t>, <full_orbital_elements_t>, <time_series_sample_t> (subprog_binary_io.cpp:44-124)
struct record_types_t
{
    h5io::Compound elements{sizeof(mh_orbital_elements)};
    h5io::Compound full{sizeof(mh_full_orbital_elements)};
    h5io::Compound sample{sizeof(time_series_sample_t)};
    record_types_t()
    {
        elements.insert_double("separation",   offsetof(mh_orbital_elements, separation));
        elements.insert_double("total_mass",   offsetof(mh_orbital_elements, total_mass));
        elements.insert_double("mass_ratio",   offsetof(mh_orbital_elements, mass_ratio));
        elements.insert_double("eccentricity", offsetof(mh_orbital_elements, eccentricity));
        full.insert_double("pomega",        offsetof(mh_full_orbital_elements, pomega));
        full.insert_double("tau",           offsetof(mh_full_orbital_elements, tau));
        full.insert_double("cm_position_x", offsetof(mh_full_orbital_elements, cm_position_x));
        full.insert_double("cm_position_y", offsetof(mh_full_orbital_elements, cm_position_y));
        full.insert_double("cm_velocity_x", offsetof(mh_full_orbital_elements, cm_velocity_x));
        full.insert_double("cm_velocity_y", offsetof(mh_full_orbital_elements, cm_velocity_y));
        full.insert("elements",             offsetof(mh_full_orbital_elements, elements), elements);
        sample.insert_double("time",                        offsetof(time_series_sample_t, time));
        sample.insert_double("disk_mass",                   offsetof(time_series_sample_t, disk_mass));
        sample.insert_double("disk_angular_momentum",       offsetof(time_series_sample_t, disk_angular_momentum));
        sample.insert_array("mass_accreted_on",             offsetof(time_series_sample_t, mass_accreted_on), 2);
        sample.insert_array("angular_momentum_accreted_on", offsetof(time_series_sample_t, angular_momentum_accreted_on), 2);
        sample.insert_array("integrated_torque_on",         offsetof(time_series_sample_t, integrated_torque_on), 2);
        sample.insert_array("work_done_on",                 offsetof(time_series_sample_t, work_done_on), 2);
        sample.insert_double("mass_ejected",                offsetof(time_series_sample_t, mass_ejected));
        sample.insert_double("angular_momentum_ejected",    offsetof(time_series_sample_t, angular_momentum_ejected));
        sample.insert("orbital_elements_acc",               offsetof(time_series_sample_t, orbital_elements_acc), full);
        sample.insert("orbital_elements_grav",              offsetof(time_series_sample_t, orbital_elements_grav), full);
        sample.insert("orbital_elements",                   offsetof(time_series_sample_t, orbital_elements), full);
        sample.insert_array("position_of_mass1",            offsetof(time_series_sample_t, position_of_mass1), 2);
        sample.insert_array("position_of_mass2",            offsetof(time_series_sample_t, position_of_mass2), 2);
    }
};

std::string format_tree_index(const mh_tree_block& b) { return h5io::format_tree_index(b.level, b.i, b.j); }

// The leaf blocks in tree order and the order in which the solver object hands cells over: one grid [n][n] (uniform depth) or
// block-major [nb][bs][bs] (graded). Files are always per block.
struct mesh_t
{
    int bs = 0, nb = 0, n = 0;
    bool graded = false;
    std::vector<mh_tree_block> blocks;
    std::vector<double> edges;                  // [nb][2][bs + 1]
    std::size_t cells() const { return std::size_t(nb) * bs * bs; }

    // q doubles per cell, solver order -> block-major
    void to_blocks(const double* sol, double* blk, int q) const
    {
        if (graded) { std::memcpy(blk, sol, cells() * q * sizeof(double)); return; }
        for (int k = 0; k < nb; ++k)
            for (int a = 0; a < bs; ++a)
                std::memcpy(blk + ((std::size_t(k) * bs + a) * bs) * q, sol + ((std::size_t(blocks[k].i) * bs + a) * n + std::size_t(blocks[k].j) * bs) * q, std::size_t(bs) * q * sizeof(double));
    }
    void from_blocks(const double* blk, double* sol, int q) const
    {
        if (graded) { std::memcpy(sol, blk, cells() * q * sizeof(double)); return; }
        for (int k = 0; k < nb; ++k)
            for (int a = 0; a < bs; ++a)
                std::memcpy(sol + ((std::size_t(blocks[k].i) * bs + a) * n + std::size_t(blocks[k].j) * bs) * q, blk + ((std::size_t(k) * bs + a) * bs) * q, std::size_t(bs) * q * sizeof(double));
    }
};

class subprog_binary : public mara::sub_program_t
{
public:
    int main(int argc, const char* argv[]) override
    {
        auto cfg = config_template().update(argc, argv);
        const std::string restart = cfg.get_string("restart");
        if (! restart.empty())
        {
            // create_run_config :155-163: template <- stored run_config <- command line
            auto file = h5io::Node::open_file(restart);
            cfg = config_template();
            h5io::read_config_into(file.open_group("run_config"), cfg);
            cfg.update(argc, argv);
        }
        cfg.pretty_print(stdout, "config");
        if (cfg.get_string("reconstruct_method") != "plm" && cfg.get_string("reconstruct_method") != "pcm")
            throw std::invalid_argument("invalid reconstruct_method '" + cfg.get_string("reconstruct_method") + "', must be plm or pcm");

        // ---- mesh: create_vertices :165-185
        mesh_t mesh;
        const int depth = cfg.get_int("depth");
        mesh.bs = cfg.get_int("block_size");
        mesh.nb = mh_binary_tree_build(mesh.bs, depth, cfg.get_double("focus_factor"), cfg.get_double("focus_index"), nullptr, 0);
        if (mesh.nb < 0) throw std::invalid_argument("binary: cannot build the block tree (block_size must be even, depth <= 12)");
        mesh.blocks.resize(mesh.nb);
        mh_binary_tree_build(mesh.bs, depth, cfg.get_double("focus_factor"), cfg.get_double("focus_index"), mesh.blocks.data(), mesh.nb);
        for (const auto& b : mesh.blocks) mesh.graded = mesh.graded || b.level != mesh.blocks[0].level;
        if (! mesh.graded && mesh.blocks[0].level != depth) throw std::invalid_argument("binary: the refinement predicate stops the tree above the requested depth");
        const int bs = mesh.bs, nb = mesh.nb;
        mesh.n = mesh.graded ? 0 : bs << depth;
        const int n = mesh.n;
        mesh.edges.resize(std::size_t(nb) * 2 * (bs + 1));
        std::vector<double> xv;
        if (mesh.graded) check(mh_binary_tree_vertices(bs, cfg.get_double("domain_radius"), mesh.blocks.data(), nb, mesh.edges.data()), "mh_binary_tree_vertices");
        else
        {
            xv.resize(n + 1);
            check(mh_binary_vertices(bs, depth, cfg.get_double("domain_radius"), xv.data()), "mh_binary_vertices");
            for (int k = 0; k < nb; ++k)
                for (int a = 0; a <= bs; ++a)
                {
                    mesh.edges[(std::size_t(k) * 2 + 0) * (bs + 1) + a] = xv[std::size_t(mesh.blocks[k].i) * bs + a];
                    mesh.edges[(std::size_t(k) * 2 + 1) * (bs + 1) + a] = xv[std::size_t(mesh.blocks[k].j) * bs + a];
                }
        }

        // ---- solver data: subprog_binary_solver_data.cpp:20-102
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
        std::vector<double> u(mesh.cells() * 3), br(mesh.cells());
        if (mesh.graded) check(mh_binary_tree_solver_data(&model, bs, mesh.blocks.data(), nb, mesh.edges.data(), u.data(), br.data(), &run.recommended_time_step), "mh_binary_tree_solver_data");
        else             check(mh_binary_solver_data(&model, n, xv.data(), xv.data(), u.data(), br.data(), &run.recommended_time_step), "mh_binary_solver_data");
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
        d.plm_theta = cfg.get_double("plm_theta");       // upstream validates reconstruct_method (solver_data.cpp:110-112) but the scheme never reads it
        {
            double min_d = 1e300;                          // smallest vertex spacing of any block: solver_data.cpp:37-52, :91
            for (int k = 0; k < nb; ++k)
                for (int ax = 0; ax < 2; ++ax)
                    for (int a = 0; a < bs; ++a)
                        min_d = std::min(min_d, mesh.edges[(std::size_t(k) * 2 + ax) * (bs + 1) + a + 1] - mesh.edges[(std::size_t(k) * 2 + ax) * (bs + 1) + a]);
            d.gst_suppr_radius = cfg.get_double("source_term_softening") * min_d;
        }

        mh_binary* solver = nullptr;
        if (mesh.graded) check(mh_binary_tree_create(&solver, cfg.get_int("device"), &d, &run, mesh.blocks.data(), nb, mesh.edges.data(), u.data(), br.data()), "mh_binary_tree_create");
        else             check(mh_binary_create(&solver, cfg.get_int("device"), &d, &run, xv.data(), xv.data(), u.data(), br.data()), "mh_binary_create");

        // ---- state: create_state :243-256
        mh_binary_state state = {};
        h5io::schedule_t schedule;
        std::vector<time_series_sample_t> time_series;          // in time order (the reference prepends to a list and writes it reversed)
        const bool qform = model.angmom_form != 0;
        std::vector<double> blk(mesh.cells() * 3);
        if (restart.empty())
        {
            state.orbital_elements.elements.total_mass = 1.0;      // create_solution :197-229
            state.orbital_elements.elements.separation = cfg.get_double("separation");
            state.orbital_elements.elements.mass_ratio = cfg.get_double("mass_ratio");
            state.orbital_elements.elements.eccentricity = cfg.get_double("eccentricity");
            check(mh_binary_set_solution(solver, nullptr, &state), "mh_binary_set_solution");
            schedule.create_and_mark_as_due("write_checkpoint");     // create_schedule :234-241
            schedule.create_and_mark_as_due("write_diagnostics");
            schedule.create_and_mark_as_due("record_time_series");
        }
        else
        {
            // mara::read<state_t> subprog_binary_io.cpp:162-192
            record_types_t types;
            auto file = h5io::Node::open_file(restart);
            auto sol = file.open_group("solution");
            int num = 0, den = 1;
            sol.read_rational("iteration", num, den);
            state.time = sol.read_double("time");
            state.iteration = num / den;
            auto tree = sol.open_group(qform ? "conserved_q" : "conserved_u");
            if (tree.names().size() != std::size_t(nb)) throw std::invalid_argument("binary: the restart file holds a different block tree");
            for (int k = 0; k < nb; ++k)
            {
                std::vector<hsize_t> shape;
                const auto cells = tree.read_cells(format_tree_index(mesh.blocks[k]), 3, shape);
                if (shape.size() != 2 || shape[0] != hsize_t(bs) || shape[1] != hsize_t(bs)) throw std::invalid_argument("binary: the restart file holds blocks of a different size");
                for (std::size_t c = 0; c < std::size_t(bs) * bs; ++c)
                    for (int q = 0; q < 3; ++q) blk[(std::size_t(k) * bs * bs + c) * 3 + q] = cells[c * 3 + (2 - q)];      // std::tuple storage order
            }
            mesh.from_blocks(blk.data(), u.data(), 3);
            sol.read_array("mass_accreted_on", state.mass_accreted_on, 2);
            sol.read_array("angular_momentum_accreted_on", state.angular_momentum_accreted_on, 2);
            sol.read_array("integrated_torque_on", state.integrated_torque_on, 2);
            sol.read_array("work_done_on", state.work_done_on, 2);
            state.mass_ejected = sol.read_double("mass_ejected");
            state.angular_momentum_ejected = sol.read_double("angular_momentum_ejected");
            sol.read_record("orbital_elements_acc", types.full, &state.orbital_elements_acc);
            sol.read_record("orbital_elements_grav", types.full, &state.orbital_elements_grav);
            sol.read_record("orbital_elements", types.full, &state.orbital_elements);
            check(mh_binary_set_solution(solver, u.data(), &state), "mh_binary_set_solution");
            time_series.resize(file.count_of("time_series"));
            file.read_records("time_series", types.sample, time_series.data());
            schedule = h5io::read_schedule(file.open_group("schedule"));
            for (const char* task : {"write_checkpoint", "write_diagnostics", "record_time_series"})
                if (! schedule.tasks.count(task)) throw std::invalid_argument(std::string("binary: the restart file's schedule lacks ") + task);
        }

        // ---- tasks: run_tasks :325-385. An interval <= 0 switches its task off (not upstream; benchmarks and tests).
        const double cpi = cfg.get_double("cpi"), dfi = cfg.get_double("dfi"), tsi = cfg.get_double("tsi");
        const bool any_task = (cpi > 0.0 || dfi > 0.0 || tsi > 0.0) && h5io::available();
        const std::string outdir = cfg.get_string("outdir");
        auto path_of = [&] (const std::string& prefix, int count)
        {
            if (! outdir.empty()) mkdir(outdir.c_str(), 0755);          // prepare_filesystem :387-391
            return (outdir.empty() ? std::string() : outdir + "/") + h5io::numbered_filename(prefix, count, "h5");
        };
        auto bodies = [&] (mh_two_body_t& B) { check(mh_two_body_state(&state.orbital_elements, state.time, &B), "mh_two_body_state"); };
        auto write_diagnostics = [&] ()
        {
            // diagnostic_fields subprog_binary_diagnostics.cpp:52-82, write<diagnostic_fields_t> subprog_binary_io.cpp:149-160
            std::vector<double> sigma(mesh.cells()), vr(mesh.cells()), vp(mesh.cells()), one(mesh.cells());
            check(mh_binary_diagnostic_fields(solver, sigma.data(), vr.data(), vp.data()), "mh_binary_diagnostic_fields");
            mh_two_body_t B;
            bodies(B);
            const std::string path = path_of("diagnostics", schedule.at("write_diagnostics").num_times_performed);
            {
                auto file = h5io::Node::create_file(path);
                h5io::write_config(file.require_group("run_config"), cfg);
                file.write("time", state.time);
                auto gv = file.require_group("vertices");
                std::vector<double> verts(std::size_t(bs + 1) * (bs + 1) * 2);
                for (int k = 0; k < nb; ++k)
                {
                    const double* xe = &mesh.edges[(std::size_t(k) * 2) * (bs + 1)];
                    const double* ye = xe + bs + 1;
                    for (int a = 0; a <= bs; ++a)
                        for (int b = 0; b <= bs; ++b) { verts[(std::size_t(a) * (bs + 1) + b) * 2] = xe[a]; verts[(std::size_t(a) * (bs + 1) + b) * 2 + 1] = ye[b]; }
                    gv.write_cells(format_tree_index(mesh.blocks[k]), {hsize_t(bs + 1), hsize_t(bs + 1)}, 2, verts.data());
                }
                const std::pair<const char*, const std::vector<double>*> fields[3] = {{"sigma", &sigma}, {"radial_velocity", &vr}, {"phi_velocity", &vp}};
                for (const auto& f : fields)
                {
                    mesh.to_blocks(f.second->data(), one.data(), 1);
                    auto g = file.require_group(f.first);
                    for (int k = 0; k < nb; ++k) g.write_grid(format_tree_index(mesh.blocks[k]), bs, bs, &one[std::size_t(k) * bs * bs]);
                }
                file.write_array("position_of_mass1", &B.body1[1], 2);
                file.write_array("position_of_mass2", &B.body2[1], 2);
            }
            std::printf("write diagnostics: %s\n", path.c_str());
            schedule.mark_as_completed("write_diagnostics");
        };
        auto record_time_series = [&] ()
        {
            time_series_sample_t s;                      // :345-366
            mh_two_body_t B;
            bodies(B);
            s.time = state.time;
            std::memcpy(s.mass_accreted_on, state.mass_accreted_on, sizeof s.mass_accreted_on);
            std::memcpy(s.angular_momentum_accreted_on, state.angular_momentum_accreted_on, sizeof s.angular_momentum_accreted_on);
            std::memcpy(s.integrated_torque_on, state.integrated_torque_on, sizeof s.integrated_torque_on);
            std::memcpy(s.work_done_on, state.work_done_on, sizeof s.work_done_on);
            s.mass_ejected = state.mass_ejected;
            s.angular_momentum_ejected = state.angular_momentum_ejected;
            check(mh_binary_disk_totals(solver, &s.disk_mass, &s.disk_angular_momentum), "mh_binary_disk_totals");
            s.orbital_elements_acc = state.orbital_elements_acc;
            s.orbital_elements_grav = state.orbital_elements_grav;
            s.orbital_elements = state.orbital_elements;
            s.position_of_mass1[0] = B.body1[1]; s.position_of_mass1[1] = B.body1[2];
            s.position_of_mass2[0] = B.body2[1]; s.position_of_mass2[1] = B.body2[2];
            time_series.push_back(s);
            schedule.mark_as_completed("record_time_series");
        };
        auto write_checkpoint = [&] ()
        {
            // :329-339: the task is marked completed first, the state written is the one that knows it
            const std::string path = path_of("chkpt", schedule.at("write_checkpoint").num_times_performed);
            schedule.mark_as_completed("write_checkpoint");
            check(mh_binary_get_solution(solver, u.data(), &state), "mh_binary_get_solution");
            mesh.to_blocks(u.data(), blk.data(), 3);
            record_types_t types;
            {
                auto file = h5io::Node::create_file(path);
                auto sol = file.require_group("solution");           // write<solution_t> subprog_binary_io.cpp:129-146
                sol.write("time", state.time);
                sol.write_rational("iteration", int(state.iteration), 1);
                auto gu = sol.require_group("conserved_u");
                auto gq = sol.require_group("conserved_q");          // the form not in use is an empty tree: an empty group
                std::vector<double> cells(std::size_t(bs) * bs * 3);
                for (int k = 0; k < nb; ++k)
                {
                    for (std::size_t c = 0; c < std::size_t(bs) * bs; ++c)
                        for (int q = 0; q < 3; ++q) cells[c * 3 + (2 - q)] = blk[(std::size_t(k) * bs * bs + c) * 3 + q];     // std::tuple storage order
                    (qform ? gq : gu).write_cells(format_tree_index(mesh.blocks[k]), {hsize_t(bs), hsize_t(bs)}, 3, cells.data());
                }
                sol.write_array("mass_accreted_on", state.mass_accreted_on, 2);
                sol.write("angular_momentum_ejected", state.angular_momentum_ejected);
                sol.write_array("integrated_torque_on", state.integrated_torque_on, 2);
                sol.write_array("work_done_on", state.work_done_on, 2);
                sol.write("mass_ejected", state.mass_ejected);
                sol.write_array("angular_momentum_accreted_on", state.angular_momentum_accreted_on, 2);
                sol.write_record("orbital_elements_acc", types.full, &state.orbital_elements_acc);
                sol.write_record("orbital_elements_grav", types.full, &state.orbital_elements_grav);
                sol.write_record("orbital_elements", types.full, &state.orbital_elements);
                h5io::write_schedule(file.require_group("schedule"), schedule);
                file.write_records("time_series", types.sample, time_series.size(), time_series.data());
                h5io::write_config(file.require_group("run_config"), cfg);
            }
            std::printf("write checkpoint: %s\n", path.c_str());
        };
        auto run_tasks = [&] ()
        {
            if (! any_task) return;
            // run_scheduled_tasks (app_schedule.hpp:161-174): which tasks run is decided before the first one does
            const bool diag = dfi > 0.0 && schedule.is_due("write_diagnostics"), series = tsi > 0.0 && schedule.is_due("record_time_series"),
                       chkpt = cpi > 0.0 && schedule.is_due("write_checkpoint");
            if (diag) write_diagnostics();
            if (series) record_time_series();
            if (chkpt) write_checkpoint();
        };
        // next_schedule :296-302, with the time of the state the step started from
        auto mark_tasks = [&] (double time)
        {
            if (! any_task) return;
            schedule.advance("write_checkpoint", time, cpi * 2 * M_PI);
            schedule.advance("write_diagnostics", time, dfi * 2 * M_PI);
            schedule.advance("record_time_series", time, tsi * 2 * M_PI);
        };

        // ---- run loop :414-438
        const double tfinal = cfg.get_double("tfinal"), cells = double(mesh.cells());
        const int batch = any_task ? 1 : std::max(1, cfg.get_int("steps_per_call"));
        const int max_iter = cfg.get_int("max_iterations");
        if (mesh.graded) std::printf("block tree: %d blocks of %d x %d zones\n", nb, bs, bs);
        if (any_task && cfg.get_int("steps_per_call") > 1) std::printf("steps_per_call ignored: tasks are scheduled every iteration (cpi, dfi, tsi <= 0 switch them off)\n");
        run_tasks();
        auto advance = [&] (int todo, bool verbose)
        {
            int safe = 0;
            const double t0 = state.time;
            const double ms = host::time_ms([&] { check(mh_binary_next(solver, todo, &safe), "mh_binary_next"); });
            check(mh_binary_get_solution(solver, nullptr, &state), "mh_binary_get_solution");
            if (safe) std::printf("negative density in updated state\n");        // what the reference prints before its safe-mode retry
            mark_tasks(t0);
            run_tasks();
            if (verbose) std::printf("[%04ld] orbits=%3.7lf kzps=%3.2lf\n", long(state.iteration), state.time / (2 * M_PI), cells * todo / ms);
            std::fflush(stdout);
        };
        while (state.time / (2 * M_PI) < tfinal && (max_iter == 0 || state.iteration < max_iter))
            advance(max_iter ? int(std::min<long>(batch, max_iter - state.iteration)) : batch, true);
        if (cfg.get_int("write_final"))
        {
            check(mh_binary_get_solution(solver, u.data(), &state), "mh_binary_get_solution");
            std::vector<double> extra;                 // grid: the vertex coordinates; tree: the block list (level, i, j as doubles); then the ten accumulators
            if (mesh.graded) for (const auto& b : mesh.blocks) { extra.push_back(b.level); extra.push_back(b.i); extra.push_back(b.j); }
            else extra = xv;
            for (double v : {state.mass_accreted_on[0], state.mass_accreted_on[1], state.angular_momentum_accreted_on[0], state.angular_momentum_accreted_on[1],
                             state.integrated_torque_on[0], state.integrated_torque_on[1], state.work_done_on[0], state.work_done_on[1],
                             state.mass_ejected, state.angular_momentum_ejected}) extra.push_back(v);
            if (mesh.graded) host::dump_state(outdir, "final.bin", {long(nb), long(bs), long(bs)}, 3, state.time, state.iteration, extra, u);
            else             host::dump_state(outdir, "final.bin", {long(n), long(n)}, 3, state.time, state.iteration, extra, u);
        }
        // upstream's closing `tasks(next(state))` (:437): one more step whose only visible effect is a task that falls due on it; final.bin
        // above is the state the loop ended with. A run cut short by max_iterations (not upstream) ends there.
        if (max_iter == 0 && any_task) advance(1, false);
        mh_binary_destroy(solver);
        return 0;
    }

    std::string name() const override { return "binary"; }
};

} // namespace

std::unique_ptr<mara::sub_program_t> make_subprog_binary() { return std::make_unique<subprog_binary>(); }
