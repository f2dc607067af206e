// `sedov` sub-program on the MI355X engine: same options, initial condition, time
// step, boundary conditions and run loop as the reference's subprog_sedov
// (src/subprog_sedov.cpp; options :55-70, vertices :366-371, IC :353-363 and
// :373-380, dt :404-405, loop :626-645, message :588-595), with the state resident
// on the device and `next_solution` (:394-421) replaced by mh_step. As upstream, the
// default system is mara::srhd and newtonian=1 selects mara::euler (:652-659).
#include <cmath>
#include <cstdio>
#include <vector>
#include "app_config.hpp"
#include "app_subprogram.hpp"
#include "host_common.hpp"
#include "h5_checkpoint.hpp"

namespace {

constexpr double gamma_law_index = 4. / 3;   // src/subprog_sedov.cpp:48
constexpr double cfl_number = 0.4;           // :49

mara::config_t config_template()
{
    return mara::config_t()
    .item("restart", "")
    .item("outdir", "data")
    .item("nr", 256)
    .item("tfinal", 1.0)
    .item("cpi", 1.0)                 // checkpoint interval (chkpt.NNNN.h5, reference layout)
    .item("tsi", 0.1)                 // time-series interval (time_series.h5, one row per sample)
    .item("dfi", 0.1)                 // diagnostics interval (diagnostics.NNNN.h5)
    .item("outer_radius", 100.0)
    .item("explosion_pressure", 1.0)
    .item("explosion_density", 1.0)
    .item("density_index", 0.0)
    .item("newtonian", 0)
    .item("device", 0);
}

class subprog_sedov : public mara::sub_program_t
{
public:
    int main(int argc, const char* argv[]) override
    {
        auto cfg = config_template().update(argc, argv);
        const std::string restart = cfg.get_string("restart");
        if (! restart.empty())
        {
            // create_run_config :466-475: template <- stored run configuration <- command line. Upstream reads the group
            // "run_config" but its checkpoints hold "config" (:492), so its own restart fails; both names are accepted here.
            auto file = h5io::Node::open_file(restart);
            cfg = config_template();
            h5io::read_config_into(file.open_group(file.has("run_config") ? "run_config" : "config"), cfg);
            cfg.update(argc, argv);
        }
        const bool newtonian = cfg.get_int("newtonian") != 0;
        cfg.pretty_print(stdout, "config");

        // vertices: 10^linspace(-0.5, log10(R), int(decades*nr)+1)
        const double decades = std::log10(cfg.get_double("outer_radius"));
        const std::size_t count = std::size_t(int(decades * cfg.get_int("nr")) + 1);
        std::vector<double> v(count);
        for (std::size_t i = 0; i < count; ++i) v[i] = std::pow(10.0, -0.5 + (decades - -0.5) * i / (count - 1));
        const std::size_t nz = count - 1;

        // initial condition: to_conserved_density(P(r_c)) * dv   (:353-363, :380; physics_euler.hpp:209-220)
        std::vector<double> u(5 * nz);
        const double dindex = cfg.get_double("density_index");
        for (std::size_t i = 0; i < nz; ++i)
        {
            const double r = (v[i] + v[i + 1]) * 0.5;
            const double d = r < 1.0 ? cfg.get_double("explosion_density") : std::pow(r, -dindex);
            const double p = r < 1.0 ? cfg.get_double("explosion_pressure") : std::pow(r, -dindex) * 1e-6;
            const double dv = (std::pow(v[i + 1], 3) - std::pow(v[i], 3)) / 3;
            double U[5];
            if (newtonian)      // mara::euler::primitive_t::to_conserved_density, physics_euler.hpp:209-220, with v = 0
            {
                const double Ue[5] = {d, d * 0.0, d * 0.0, d * 0.0, 0.5 * d * (0.0 * 0.0 + 0.0 * 0.0 + 0.0 * 0.0) + p / (gamma_law_index - 1)};
                for (int q = 0; q < 5; ++q) U[q] = Ue[q];
            }
            else                // mara::srhd::primitive_t::to_conserved_density, physics_srhd.hpp:213-227, with u = 0
            {
                const double W = std::sqrt(1.0 + (0.0 * 0.0 + 0.0 * 0.0 + 0.0 * 0.0));
                const double h = (d + p * (1.0 + 1.0 / (gamma_law_index - 1.0))) / d;
                const double D = d * W;
                const double Us[5] = {D, D * 0.0 * h, D * 0.0 * h, D * 0.0 * h, D * h * W - p - D};
                for (int q = 0; q < 5; ++q) U[q] = Us[q];
            }
            for (int q = 0; q < 5; ++q) u[5 * i + q] = U[q] * dv;
        }

        mh_ctx* ctx = nullptr;
        host::check(mh_create(&ctx, cfg.get_int("device")), nullptr, "mh_create");
        mh_sedov_desc d = {int(nz), gamma_law_index, newtonian ? MH_SYSTEM_EULER : MH_SYSTEM_SRHD, MH_ARITH_STRICT};
        host::check(mh_sedov_configure(ctx, &d, v.data()), ctx, "mh_sedov_configure");
        host::check(mh_upload(ctx, u.data(), nz), ctx, "mh_upload");

        const double dt = cfl_number * (v[1] - v[0]);
        const double tfinal = cfg.get_double("tfinal");
        double time = 0.0;
        long iteration = 0;
        h5io::schedule_t schedule;
        const char* task_names[3] = {"write_checkpoint", "write_diagnostics", "write_time_series"};
        if (restart.empty())
        {
            for (const char* task : task_names) schedule.create_and_mark_as_due(task);        // new_schedule :427-434
        }
        else
        {
            auto file = h5io::Node::open_file(restart);
            auto sol = file.open_group("solution");                    // read_solution :337-345
            int num = 0, den = 1;
            sol.read_rational("iteration", num, den);
            time = sol.read_double("time");
            iteration = num / den;
            if (sol.read_vector("vertices") != v) throw std::invalid_argument("sedov: the restart file's vertices differ from this configuration's");
            std::vector<hsize_t> shape;
            u = sol.read_cells("conserved", 5, shape);
            if (shape.size() != 1 || shape[0] != nz) throw std::invalid_argument("sedov: the restart file holds a different number of zones");
            host::check(mh_upload(ctx, u.data(), nz), ctx, "mh_upload");
            schedule = h5io::read_schedule(file.open_group("schedule"));
            for (const char* task : task_names) if (! schedule.tasks.count(task)) schedule.create_and_mark_as_due(task);
        }
        const std::string outdir = cfg.get_string("outdir");
        const std::string prefix = outdir.empty() ? std::string() : outdir + "/";
        const double cpi = cfg.get_double("cpi"), dfi = cfg.get_double("dfi"), tsi = cfg.get_double("tsi");
        // a task interval <= 0 switches that task off (not upstream: tests and benchmarks)
        const bool tasks_on = (cpi > 0.0 || dfi > 0.0 || tsi > 0.0) && h5io::available();
        const char* columns[6] = {"time", "shock_radius", "shock_radius_upstream", "shock_radius_downstream", "shock_radius_interpolated", "shock_velocity"};
        if (tasks_on && ! outdir.empty()) mkdir(outdir.c_str(), 0755);
        struct stat st;
        if (tasks_on && tsi > 0.0 && (restart.empty() || stat((prefix + "time_series.h5").c_str(), &st) != 0))
        {
            // prepare_filesystem :596-615: time_series.h5 with one empty, extendible dataset per column (chunks of 1000) and the run_config.
            // Upstream creates it for fresh runs only, so a run restarted into another outdir dies at its first sample; here the file
            // is created there too (its rows before the restart stay at the fill value).
            auto file = h5io::Node::create_file(prefix + "time_series.h5");
            for (const char* name : columns) file.create_unlimited(name, 1000);
            h5io::write_config(file.require_group("run_config"), cfg);
        }
        // make_diagnostic_fields / compute_time_series_data :252-308: the fields and the shock locator come from the device, the six
        // scalars are a few operations on them (parabola_vertex math_polynomial.hpp:206-215, solve_for_shock_velocity :96-114)
        std::vector<double> fields(4 * nz);
        auto time_series_data = [&] (double series[6])
        {
            int32_t idx[3];
            host::check(mh_sedov_diagnostics(ctx, fields.data(), idx), ctx, "mh_sedov_diagnostics");
            const std::size_t shock = idx[0], down = idx[1], up = idx[2];
            const double* pr = &fields[nz]; (void) pr;
            const double* dn = &fields[2 * nz];
            const double* vc = &fields[3 * nz];
            auto rc = [&] (std::size_t i) { return (v[i] + v[i + 1]) * 0.5; };
            series[0] = time;
            series[1] = v[shock];
            series[2] = rc(up);
            series[3] = rc(down);
            if (down >= 1 && down + 1 < nz)
            {
                const double x1 = rc(down - 1), x2 = rc(down), x3 = rc(down + 1), y1 = vc[down - 1], y2 = vc[down], y3 = vc[down + 1];
                const double d = (x1 - x2) * (x1 - x3) * (x2 - x3);
                const double A = (x3 * (y2 - y1) + x2 * (y1 - y3) + x1 * (y3 - y2)) / d;
                const double B = (x3 * x3 * (y1 - y2) + x2 * x2 * (y3 - y1) + x1 * x1 * (y2 - y3)) / d;
                series[4] = -B / (2 * A);
            }
            else series[4] = std::nan("");        // upstream indexes outside the array here
            const double d1 = dn[up], d2 = dn[down], u1 = vc[up], u2 = vc[down];
            if (newtonian) series[5] = (d2 * u2 - d1 * u1) / (d2 - d1);
            else
            {
                const double g1 = std::sqrt(1.0 + (u1 * u1 + 0.0 * 0.0 + 0.0 * 0.0)), g2 = std::sqrt(1.0 + (u2 * u2 + 0.0 * 0.0 + 0.0 * 0.0));
                series[5] = (d2 * u2 - d1 * u1) / (d2 * g2 - d1 * g1);
            }
        };
        auto write_checkpoint = [&] ()
        {
            // write_checkpoint :486-495, write_solution :329-335
            host::check(mh_download(ctx, u.data(), nz), ctx, "mh_download");
            const std::string path = prefix + h5io::numbered_filename("chkpt", schedule.at("write_checkpoint").num_times_performed, "h5");
            {
                auto file = h5io::Node::create_file(path);
                auto sol = file.require_group("solution");
                sol.write("time", time);
                sol.write_rational("iteration", int(iteration), 1);
                sol.write("vertices", v);
                sol.write_cells("conserved", {hsize_t(nz)}, 5, u.data());
                h5io::write_schedule(file.require_group("schedule"), schedule);
                h5io::write_config(file.require_group("config"), cfg);
            }
            std::printf("write checkpoint: %s\n", path.c_str());
        };
        auto write_diagnostics = [&] ()
        {
            // write_diagnostics :497-514
            double series[6];
            time_series_data(series);
            const std::string path = prefix + h5io::numbered_filename("diagnostics", schedule.at("write_diagnostics").num_times_performed, "h5");
            {
                auto file = h5io::Node::create_file(path);
                auto column = [&] (int k) { return std::vector<double>(fields.begin() + k * nz, fields.begin() + (k + 1) * nz); };
                std::vector<double> rcs(nz);
                for (std::size_t i = 0; i < nz; ++i) rcs[i] = (v[i] + v[i + 1]) * 0.5;
                file.write("gas_pressure", column(1));
                file.write("mass_density", column(2));
                file.write("specific_entropy", column(0));
                file.write("radial_gamma_beta", column(3));
                file.write("radial_coordinates", rcs);
                for (int k = 0; k < 6; ++k) file.write(columns[k], series[k]);
            }
            std::printf("write diagnostics: %s\n", path.c_str());
        };
        auto write_time_series = [&] ()
        {
            // write_time_series :516-529: row number = how often the task has run
            double series[6];
            time_series_data(series);
            auto file = h5io::Node::open_file_rw(prefix + "time_series.h5");
            const hsize_t row = hsize_t(schedule.at("write_time_series").num_times_performed);
            for (int k = 0; k < 6; ++k) file.append(columns[k], row, series[k]);
        };
        // run_tasks :559-583: which tasks run is read from the incoming schedule
        auto run_tasks = [&] ()
        {
            if (! tasks_on) return;
            const bool chk = cpi > 0.0 && schedule.is_due("write_checkpoint"), diag = dfi > 0.0 && schedule.is_due("write_diagnostics"),
                       series = tsi > 0.0 && schedule.is_due("write_time_series");
            if (chk) { write_checkpoint(); schedule.mark_as_completed("write_checkpoint"); }
            if (diag) { write_diagnostics(); schedule.mark_as_completed("write_diagnostics"); }
            if (series) { write_time_series(); schedule.mark_as_completed("write_time_series"); }
        };
        run_tasks();

        // one `run_tasks(next(state))` (:542-549, :559-583)
        auto advance = [&] (bool verbose)
        {
            const double ms = host::time_ms([&] {
                host::check(mh_step(ctx, dt, 1), ctx, "mh_step");
                host::check(mh_synchronize(ctx), ctx, "mh_synchronize");
            });
            if (tasks_on)
            {
                // next_schedule :445-457 looks at the time of the state the step STARTED from (SedovProblem::next :542-549)
                schedule.advance("write_checkpoint", time, cpi);
                schedule.advance("write_diagnostics", time, dfi);
                schedule.advance("write_time_series", time, tsi);
            }
            time += dt;
            iteration += 1;
            run_tasks();
            // the reference throws out of the failing step (recover_primitive, physics_srhd.hpp:430-449): every bit, every step
            host::throw_on_status(ctx, newtonian ? MH_SYSTEM_EULER : MH_SYSTEM_SRHD);
            if (verbose && iteration % 100 == 0)
            {
                std::printf("[%04ld] t=%3.7lf kzps=%3.2lf\n", iteration, time, count / ms);   // counts vertices, like the reference (:592)
            }
        };
        while (time < tfinal) advance(true);
        host::check(mh_download(ctx, u.data(), nz), ctx, "mh_download");
        host::dump_state(cfg.get_string("outdir"), "final.bin", {long(nz)}, 5, time, iteration, v, u);
        // upstream's closing `run_tasks_on_next(state)` (:644): one more step whose only visible effect is a task that falls due on it;
        // final.bin above is the state the loop ended with
        if (tasks_on) advance(false);
        mh_destroy(ctx);
        return 0;
    }

    std::string name() const override { return "sedov"; }
};

} // namespace

std::unique_ptr<mara::sub_program_t> make_subprog_sedov() { return std::make_unique<subprog_sedov>(); }
