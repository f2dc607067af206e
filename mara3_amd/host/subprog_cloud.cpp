// `cloud` sub-program on the MI355X engine: 2-D axisymmetric jet-cloud interaction in spherical-polar
// coordinates, special-relativistic (mara::srhd). Same options, grid, units, initial condition, time step,
// nozzle boundary condition, run loop and iteration message as the reference's subprog_cloud
// (src/subprog_cloud.cpp: options :60-87, units :319-332, IC and grid :610-662, dt :678-679, nozzle
// :466-493, loop :916-936, message :856-861). CloudProblem::advance / next_solution (:511-584, :676-697)
// are replaced by mh_step; the state stays on the device between tasks. Checkpoint / diagnostics / time
// series (HDF5) are out of scope this round: the final state is dumped as raw binary.
#include <cmath>
#include <cstdio>
#include <vector>
#include "app_config.hpp"
#include "app_subprogram.hpp"
#include "host_common.hpp"
#include "h5_checkpoint.hpp"
#include "models.hpp"

namespace {

constexpr double gamma_law_index = 4. / 3;      // src/subprog_cloud.cpp:52
constexpr double light_speed_cgs = 2.998e10;    // :53
constexpr double solar_mass_cgs = 1.989e33;     // :54

mara::config_t config_template()
{
    return mara::config_t()
    .item("restart", "")
    .item("outdir", "data")
    .item("nr", 256)
    .item("tfinal", 1.0)
    .item("cpi", 10.0)                // checkpoint interval (chkpt.NNNN.h5, reference layout)
    .item("tsi", 0.1)                 // time-series interval: the task exists upstream but writes nothing (:801-803)
    .item("dfi", 1.0)                 // diagnostics interval (diagnostics.NNNN.h5, make_diagnostic_fields evaluated on the device)
    .item("num_decades", 2.0)
    .item("inner_radius", 3e08)
    .item("cloud_cutoff", 3e10)
    .item("cloud_mass", 2e-2)
    .item("density_index", 2.0)
    .item("density_index2", 6.0)
    .item("jet_delay_time", 1.0)
    .item("jet_total_energy", 1e50)
    .item("jet_duration", 1.0)
    .item("jet_gamma_beta", 10.0)
    .item("jet_opening_angle", 0.1)
    .item("jet_structure_exp", 2.0)
    .item("cfl_number", 0.4)
    .item("rk_order", 1)
    .item("reconstruct_method", 2)
    .item("plm_theta", 1.2)
    .item("temperature_floor", 1e-8)
    .item("arith", "strict")         // strict (bit-identical to the reference) | fast (L1 <= 1e-12); not a reference option
    .item("fuse", 0)                 // the RK2 step as one launch (arith=fast, PLM): 0 = where available, -1 = never, 1 = required; not a reference option
    .item("planar", 0)               // skip the azimuthal momentum where field and nozzle row have none (verified): 0 = auto, -1 = never; not a reference option
    .item("chunk_rows", 0)           // rows marched per wave (0 = the library's default); not a reference option
    .item("profile", 0)              // print the average stage-kernel time from HIP events at the end; not a reference option
    .item("max_steps", 0)            // stop after this many steps (0 = run to tfinal); not a reference option
    .item("write_inflow", 0)         // also dump the nozzle row of the first step (tests)
    .item("device", 0)
    .item("gpus", 1);                // radial slabs over this many GPUs of the node, ONE process (the reference's `evaluate_on<N>` thread slabs,
                                     // :527-581, with a device per slab); more slabs than visible devices share them round-robin. Not a reference option
}

// conserved density of a primitive state, mara::srhd::primitive_t::to_conserved_density (src/physics_srhd.hpp:213-227);
// host-side because the initial condition is built on the host, as in the reference (:658)
void to_conserved_density(const double P[5], double U[5])
{
    const double W = std::sqrt(1.0 + (P[1] * P[1] + P[2] * P[2] + P[3] * P[3]));
    const double h = (P[0] + P[4] * (1.0 + 1.0 / (gamma_law_index - 1.0))) / P[0];
    const double D = P[0] * W;
    U[0] = D;
    U[1] = D * P[1] * h;
    U[2] = D * P[2] * h;
    U[3] = D * P[3] * h;
    U[4] = D * h * W - P[4] - D;
}

class subprog_cloud : public mara::sub_program_t
{
public:
    int main(int argc, const char* argv[]) override
    {
        auto cfg = config_template().update(argc, argv);
        const std::string restart = cfg.get_string("restart");
        if (! restart.empty())
        {
            // create_run_config :743-752: template <- stored "config" group <- command line
            auto file = h5io::Node::open_file(restart);
            cfg = config_template();
            h5io::read_config_into(file.open_group("config"), cfg);
            cfg.update(argc, argv);
        }
        cfg.pretty_print(stdout, "config");

        model::EjectaParams ejecta;
        ejecta.cloud_slope = cfg.get_double("density_index");
        model::HaloParams halo;
        halo.reference_radius = cfg.get_double("inner_radius");
        halo.break_radius = cfg.get_double("cloud_cutoff");
        halo.inner_slope = cfg.get_double("density_index");
        halo.outer_slope = cfg.get_double("density_index2");
        halo = model::halo_scaled_to_mass(halo, cfg.get_double("cloud_mass") * solar_mass_cgs);
        model::EngineParams engine;
        engine.base_radius = cfg.get_double("inner_radius");
        engine.energy = cfg.get_double("jet_total_energy");
        engine.duration = cfg.get_double("jet_duration");
        engine.angular_exponent = cfg.get_double("jet_structure_exp");
        engine.opening_angle = cfg.get_double("jet_opening_angle");
        engine.four_velocity0 = cfg.get_double("jet_gamma_beta");

        // reference units (:319-332): length = r0, mass = total atmosphere mass, time = r0 / c
        const double ref_length = halo.reference_radius;
        const double ref_mass = model::halo_mass(halo);
        const double ref_time = halo.reference_radius / light_speed_cgs;
        const double ref_density = ref_mass / std::pow(ref_length, 3);

        // grid (:645-651): r = 10^linspace(0, decades, int(decades*nr)+1), theta = linspace(0, pi, nr+1)
        const int nr_opt = cfg.get_int("nr");
        const double num_decades = cfg.get_double("num_decades");
        const std::size_t nrv = std::size_t(int(num_decades * nr_opt) + 1), nqv = std::size_t(nr_opt + 1);
        std::vector<double> rv(nrv), qv(nqv);
        for (std::size_t i = 0; i < nrv; ++i) rv[i] = std::pow(10.0, 0.0 + (num_decades - 0.0) * i / (nrv - 1));
        for (std::size_t j = 0; j < nqv; ++j) qv[j] = 0.0 + (M_PI - 0.0) * j / (nqv - 1);
        const int nr = int(nrv - 1), nq = int(nqv - 1);

        // initial condition (:626-660): primitive from the envelop model at the centroid radius, times the cell volume
        std::vector<double> u(std::size_t(5) * nr * nq);
        const double jet_delay_time = cfg.get_double("jet_delay_time");
        for (int i = 0; i < nr; ++i)
        {
            const double rc = (rv[i] + rv[i + 1]) * 0.5;
            const double r_cm = rc * ref_length;
            const model::EjectaState shell = model::ejecta_at(ejecta, r_cm, jet_delay_time);
            const double density = shell.density / ref_density;
            const double gamma_beta = shell.four_velocity;
            const double P[5] = {density, gamma_beta, 0.0, 0.0, density * 1e-6};
            double U[5];
            to_conserved_density(P, U);
            const double d3 = rv[i + 1] * rv[i + 1] * rv[i + 1] - rv[i] * rv[i] * rv[i];
            for (int j = 0; j < nq; ++j)
            {
                const double dmj = -std::cos(qv[j + 1]) - -std::cos(qv[j]);
                const double dv = ((d3 + d3) * 0.5) * ((dmj + dmj) * 0.5) * 2 * M_PI / 3.0;      // cell_volumes (:276-283)
                for (int q = 0; q < 5; ++q) u[(std::size_t(i) * nq + j) * 5 + q] = U[q] * dv;
            }
        }

        mh_cloud_desc d = {};
        d.nr = nr; d.nq = nq; d.nr_global = nr; d.row_offset = 0;
        d.gamma = gamma_law_index;
        d.plm_theta = cfg.get_int("reconstruct_method") == 1 ? -1.0 : cfg.get_double("plm_theta");
        if (cfg.get_int("reconstruct_method") != 1 && cfg.get_int("reconstruct_method") != 2) throw std::invalid_argument("reconstruct_method must be 1 or 2");
        d.temperature_floor = cfg.get_double("temperature_floor");
        d.bc_lo0 = MH_BC_INFLOW; d.bc_hi0 = MH_BC_OUTFLOW;
        d.arith = cfg.get_string("arith") == "fast" ? MH_ARITH_FAST : MH_ARITH_STRICT;
        d.fuse_stages = int(cfg.get_int("fuse"));
        d.chunk_rows = int(cfg.get_int("chunk_rows"));
        d.planar = int(cfg.get_int("planar"));

        mh_ctx* ctx = nullptr;
        host::check(mh_create(&ctx, cfg.get_int("device")), nullptr, "mh_create");
        host::check(mh_cloud_configure(ctx, &d, rv.data(), qv.data(), cfg.get_int("rk_order")), ctx, "mh_cloud_configure");
        // gpus > 1: the step runs on a group of radial slabs (BASELINE config 4: slab decomposition + two-row halo per stage), member r on
        // device r of the node; the context above then only serves make_diagnostic_fields when that task is due
        const int gpus = int(cfg.get_int("gpus"));
        if (gpus < 1 || gpus > 64) throw std::invalid_argument("gpus must be 1..64");
        std::vector<mh_slab*> slabs(gpus > 1 ? gpus : 0, nullptr);
        if (gpus > 1)
        {
            const int visible = mh_device_count();
            std::vector<int> ids(gpus);
            for (int r = 0; r < gpus; ++r) ids[r] = r % (visible > 0 ? visible : 1);
            if (visible < gpus) std::printf("gpus=%d on %d visible device(s): slabs share devices round-robin\n", gpus, visible);
            if (mh_slab_cloud_group_create_on(slabs.data(), &d, rv.data(), qv.data(), int(cfg.get_int("rk_order")), gpus, ids.data()) != MH_OK)
                throw std::runtime_error(std::string("mh_slab_cloud_group_create_on: ") + mh_last_error(nullptr));
            // (1: the one-launch RK2 step across the radial cuts - four ghost rows, one exchange per step; round 5)
            std::printf("slab launches per step:");
            for (mh_slab* sl : slabs) std::printf(" %d", mh_slab_launches_per_step(sl));
            std::printf("\n");
        }
        auto group_check = [] (int rc, const char* what) { if (rc != MH_OK) throw std::runtime_error(std::string(what) + ": " + mh_last_error(nullptr)); };
        auto upload_solution = [&] ()
        {
            if (gpus > 1) group_check(mh_slab_group_upload(slabs.data(), gpus, u.data()), "mh_slab_group_upload");
            else          host::check(mh_upload(ctx, u.data(), std::size_t(nr) * nq), ctx, "mh_upload");
        };
        auto download_solution = [&] ()
        {
            if (gpus > 1) group_check(mh_slab_group_download(slabs.data(), gpus, u.data()), "mh_slab_group_download");
            else          host::check(mh_download(ctx, u.data(), std::size_t(nr) * nq), ctx, "mh_download");
        };
        upload_solution();

        const double dt = (rv[1] - rv[0]) / 1.0 * cfg.get_double("cfl_number");
        const double tfinal = cfg.get_double("tfinal");
        const long max_steps = cfg.get_int("max_steps");
        double time = 0.0;
        long iteration = 0;
        std::vector<double> inflow(std::size_t(5) * nq, 0.0), inflow_first;
        double timed_ms = 0.0;               // profile = 1: the steps after the first 30 (the GPU's clocks settle over the first ~25 launches)
        long timed_steps = 0;
        h5io::schedule_t schedule;
        if (restart.empty())
        {
            schedule.create_and_mark_as_due("write_checkpoint");        // new_schedule :703-710
            schedule.create_and_mark_as_due("write_diagnostics");
            schedule.create_and_mark_as_due("write_time_series");
        }
        else
        {
            auto file = h5io::Node::open_file(restart);
            auto sol = file.open_group("solution");                    // read_solution :599-608
            int num = 0, den = 1;
            sol.read_rational("iteration", num, den);
            time = sol.read_double("time");
            iteration = num / den;
            if (sol.read_vector("radial_vertices") != rv || sol.read_vector("polar_vertices") != qv)
                throw std::invalid_argument("cloud: the restart file's vertices differ from this configuration's");
            std::vector<hsize_t> shape;
            u = sol.read_cells("conserved", 5, shape);
            if (shape.size() != 2 || shape[0] != hsize_t(nr) || shape[1] != hsize_t(nq)) throw std::invalid_argument("cloud: the restart file holds a different grid");
            upload_solution();
            schedule = h5io::read_schedule(file.open_group("schedule"));
            for (const char* task : {"write_checkpoint", "write_diagnostics", "write_time_series"})
                if (! schedule.tasks.count(task)) schedule.create_and_mark_as_due(task);
        }
        const std::string outdir = cfg.get_string("outdir");
        const double units[3] = {ref_length, ref_mass, ref_time};
        auto path_of = [&] (const char* prefix, int count)
        {
            if (! outdir.empty()) mkdir(outdir.c_str(), 0755);
            return (outdir.empty() ? std::string() : outdir + "/") + h5io::numbered_filename(prefix, count, "h5");
        };
        auto write_checkpoint = [&] ()
        {
            // write_checkpoint :758-767, write_solution :590-597
            download_solution();
            const std::string path = path_of("chkpt", schedule.at("write_checkpoint").num_times_performed);
            {
                auto file = h5io::Node::create_file(path);
                auto sol = file.require_group("solution");
                sol.write("time", time);
                sol.write_rational("iteration", int(iteration), 1);
                sol.write("radial_vertices", rv);
                sol.write("polar_vertices", qv);
                sol.write_cells("conserved", {hsize_t(nr), hsize_t(nq)}, 5, u.data());
                h5io::write_schedule(file.require_group("schedule"), schedule);
                h5io::write_config(file.require_group("config"), cfg);
            }
            std::printf("write checkpoint: %s\n", path.c_str());
        };
        auto write_diagnostics = [&] ()
        {
            // write_diagnostics :769-799: make_diagnostic_fields (:334-433) is evaluated on the device, the file gets its results
            std::vector<double> fields(std::size_t(5) * nr * nq), columns(std::size_t(15) * nq);
            if (gpus > 1)          // gather the slabs into the context (one device) for the diagnostic fields: only when the task is due
            {
                download_solution();
                host::check(mh_upload(ctx, u.data(), std::size_t(nr) * nq), ctx, "mh_upload");
            }
            host::check(mh_cloud_diagnostics(ctx, units, fields.data(), columns.data()), ctx, "mh_cloud_diagnostics");
            const std::string path = path_of("diagnostics", schedule.at("write_diagnostics").num_times_performed);
            {
                auto file = h5io::Node::create_file(path);
                const std::size_t plane = std::size_t(nr) * nq;
                file.write("time", time * ref_time);
                file.write_grid("gas_pressure", nr, nq, &fields[1 * plane]);
                file.write_grid("mass_density", nr, nq, &fields[0 * plane]);
                file.write_grid("specific_entropy", nr, nq, &fields[2 * plane]);
                file.write_grid("radial_energy_flow", nr, nq, &fields[4 * plane]);
                file.write_grid("radial_gamma_beta", nr, nq, &fields[3 * plane]);
                std::vector<double> rv_cm(rv);
                for (auto& r : rv_cm) r = r * ref_length;
                file.write("radial_vertices", rv_cm);
                file.write("polar_vertices", qv);
                const char* names[15] = {"total_energy_at_theta", "solid_angle_at_theta", "shock_midpoint_radius", "shock_upstream_radius",
                                         "shock_pressure_radius", "shock_luminosity_radius", "postshock_flow_gamma", "postshock_flow_power",
                                         "postshock_flow_power02", "postshock_flow_power04", "postshock_flow_power08", "postshock_flow_power16",
                                         "postshock_flow_power32", "postshock_flow_power64", "postshock_flow_power_max"};
                for (int k = 0; k < 15; ++k) file.write(names[k], std::vector<double>(columns.begin() + std::size_t(k) * nq, columns.begin() + std::size_t(k + 1) * nq));
            }
            std::printf("write diagnostics: %s\n", path.c_str());
        };
        // run_tasks :826-849: which tasks run is read from the incoming schedule; a task interval <= 0 switches that task off
        // (not upstream: tests and benchmarks)
        const bool tasks_on = h5io::available();
        const double cpi = cfg.get_double("cpi"), dfi = cfg.get_double("dfi"), tsi = cfg.get_double("tsi");
        auto run_tasks = [&] ()
        {
            if (! tasks_on) return;
            const bool chk = cpi > 0.0 && schedule.is_due("write_checkpoint"), diag = dfi > 0.0 && schedule.is_due("write_diagnostics"),
                       series = tsi > 0.0 && schedule.is_due("write_time_series");
            if (chk) { write_checkpoint(); schedule.mark_as_completed("write_checkpoint"); }
            if (diag) { write_diagnostics(); schedule.mark_as_completed("write_diagnostics"); }
            if (series) schedule.mark_as_completed("write_time_series");                 // write_time_series is empty upstream (:801-803)
        };
        run_tasks();

        // one `run_tasks(next(state))` (:811-849)
        auto advance = [&] (bool verbose)
        {
            // nozzle row at the step-start time (:466-493); both RK stages use it (:524)
            const double t_seconds = time * ref_time;
            for (int j = 0; j < nq; ++j)
            {
                const double q = (qv[j] + qv[j + 1]) * 0.5;
                inflow[5 * j + 0] = model::engine_base_density(engine) / ref_density;
                inflow[5 * j + 1] = model::engine_four_velocity(engine, q, t_seconds) + model::engine_four_velocity(engine, M_PI - q, t_seconds);
            }
            if (iteration == 0) inflow_first = inflow;
            const double ms = host::time_ms([&] {
                if (gpus > 1)
                {
                    group_check(mh_slab_group_set_inflow(slabs.data(), gpus, inflow.data()), "mh_slab_group_set_inflow");
                    group_check(mh_slab_group_step(slabs.data(), gpus, dt, 1), "mh_slab_group_step");
                    for (mh_slab* sl : slabs) group_check(mh_slab_synchronize(sl), "mh_slab_synchronize");
                }
                else
                {
                    host::check(mh_cloud_set_inflow(ctx, inflow.data()), ctx, "mh_cloud_set_inflow");
                    host::check(mh_step(ctx, dt, 1), ctx, "mh_step");
                    host::check(mh_synchronize(ctx), ctx, "mh_synchronize");
                }
            });
            if (tasks_on)
            {
                // next_schedule :720-733 looks at the time of the state the step STARTED from (CloudProblem::next :811-817)
                schedule.advance("write_checkpoint", time, cpi);
                schedule.advance("write_diagnostics", time, dfi);
                schedule.advance("write_time_series", time, tsi);
            }
            time += dt;
            iteration += 1;
            run_tasks();
            // the reference throws out of the failing step (physics_srhd.hpp:430-449)
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
                host::throw_on_result(worst, MH_SYSTEM_SRHD);
            }
            else host::throw_on_status(ctx, MH_SYSTEM_SRHD);
            if (verbose) std::printf("[%04ld] t=%3.7lf kzps=%3.2lf\n", iteration, time, double(nrv) * nqv / ms);    // vertices, like the reference (:858)
            if (iteration > 30) { timed_ms += ms; timed_steps += 1; }
        };
        while (time < tfinal && (max_steps == 0 || iteration < max_steps)) advance(true);
        download_solution();
        std::vector<double> vertices(rv);
        vertices.insert(vertices.end(), qv.begin(), qv.end());
        host::dump_state(cfg.get_string("outdir"), "final.bin", {long(nr), long(nq)}, 5, time, iteration, vertices, u);
        if (cfg.get_int("write_inflow"))
            host::dump_state(cfg.get_string("outdir"), "inflow0.bin", {long(nq)}, 5, 0.0, 0, {}, inflow_first);
        // upstream's closing `run_tasks_on_next(state)` (:935): one more step whose only visible effect is a task that falls due on it.
        // final.bin above is the state the loop ended with; a run cut short by max_steps (not upstream) ends there.
        if (max_steps == 0 && tasks_on) advance(false);
        if (cfg.get_int("profile") && timed_steps > 0)
        {
            // one JSON line of this process's own timing (kept beside the rocprofv3 trace of the same run: profiles/r05, tests/test_profiles_cpu.py)
            int lps = gpus > 1 ? mh_slab_launches_per_step(slabs[0]) : 0;
            std::printf("{\"run\": \"mara_hip cloud nr=%d gpus=%d arith=%s\", \"ms_per_step\": %.6f, \"steps_timed\": %ld, \"zones\": %ld, \"value\": %.3f, \"unit\": \"Mcells/s\", "
                        "\"slab_launches_per_step\": %d, \"note\": \"host-timed steps 31.. (nozzle row evaluated on the host and uploaded, step, synchronise)\"}\n",
                        nr, gpus, cfg.get_string("arith").c_str(), timed_ms / timed_steps, timed_steps, long(nr) * nq, double(nr) * nq / (timed_ms / timed_steps) / 1e3, lps);
        }
        // (profile = 1 steps the live solution further: only now, after every task of the run - the closing one included - has seen its state)
        if (cfg.get_int("profile") && gpus == 1)
        {
            std::printf("step kernels: %s\n", mh_field_is_planar(ctx) ? "planar (no azimuthal momentum in field and nozzle row: verified)" : "general");
            // after the run and its output: five further steps in ONE call, i.e. between ONE pair of events (events around every step put two markers
            // between consecutive kernels and read long on sub-millisecond launches); the nozzle row stays that of the last step
            const int extra = 5;
            double avg_ms = 0.0;
            int launches = 0;
            host::check(mh_step(ctx, dt, 40), ctx, "mh_step");                      // lead-in: the download and the file above left the GPU idle for seconds,
                                                                                    // and the first ~25 launches after an idle period run up to twice as long
            host::check(mh_profile_enable(ctx, 1), ctx, "mh_profile_enable");
            host::check(mh_step(ctx, dt, extra), ctx, "mh_step");
            host::check(mh_profile_read(ctx, &avg_ms, &launches), ctx, "mh_profile_read");
            host::check(mh_profile_enable(ctx, 0), ctx, "mh_profile_enable");
            std::printf("profile: stage kernel avg %.6f ms over %d launches (%d per step; one pair of events around %d further steps)\n", avg_ms, launches, launches / extra, extra);
            int32_t word = 0;
            host::check(mh_status_word(ctx, &word), ctx, "mh_status_word");       // (cleared: these steps are not part of the run)
        }
        for (mh_slab* sl : slabs) mh_slab_destroy(sl);
        mh_destroy(ctx);
        return 0;
    }

    std::string name() const override { return "cloud"; }
};

} // namespace

std::unique_ptr<mara::sub_program_t> make_subprog_cloud() { return std::make_unique<subprog_cloud>(); }
