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
    .item("cpi", 1.0)                 // checkpoint interval (chkpt.NNNN.h5, reference layout); tsi / dfi tasks are out of scope
    .item("tsi", 0.1)
    .item("dfi", 0.1)
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
        if (restart.empty())
        {
            schedule.create_and_mark_as_due("write_checkpoint");        // new_schedule :427-434
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
            if (! schedule.tasks.count("write_checkpoint")) schedule.create_and_mark_as_due("write_checkpoint");
        }
        const std::string outdir = cfg.get_string("outdir");
        auto run_tasks = [&] ()
        {
            if (! schedule.is_due("write_checkpoint")) return;
            // write_checkpoint :486-495, write_solution :329-335
            host::check(mh_download(ctx, u.data(), nz), ctx, "mh_download");
            if (! outdir.empty()) mkdir(outdir.c_str(), 0755);
            const std::string path = (outdir.empty() ? std::string() : outdir + "/") + h5io::numbered_filename("chkpt", schedule.at("write_checkpoint").num_times_performed, "h5");
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
            schedule.mark_as_completed("write_checkpoint");
        };
        const bool checkpoints = cfg.get_double("cpi") > 0.0 && h5io::available();        // cpi <= 0 switches the task off (not upstream: tests and benchmarks)
        if (checkpoints) run_tasks();

        while (time < tfinal)
        {
            const double ms = host::time_ms([&] {
                host::check(mh_step(ctx, dt, 1), ctx, "mh_step");
                host::check(mh_synchronize(ctx), ctx, "mh_synchronize");
            });
            // next_schedule :445-457 looks at the time of the state the step STARTED from (SedovProblem::next :542-549)
            if (checkpoints) schedule.advance("write_checkpoint", time, cfg.get_double("cpi"));
            time += dt;
            iteration += 1;
            if (checkpoints) run_tasks();
            if (iteration % 100 == 0)
            {
                host::throw_on_status(ctx);
                std::printf("[%04ld] t=%3.7lf kzps=%3.2lf\n", iteration, time, count / ms);   // counts vertices, like the reference (:592)
            }
        }
        host::throw_on_status(ctx);
        host::check(mh_download(ctx, u.data(), nz), ctx, "mh_download");
        host::dump_state(cfg.get_string("outdir"), "final.bin", {long(nz)}, 5, time, iteration, v, u);
        mh_destroy(ctx);
        return 0;
    }

    std::string name() const override { return "sedov"; }
};

} // namespace

std::unique_ptr<mara::sub_program_t> make_subprog_sedov() { return std::make_unique<subprog_sedov>(); }
