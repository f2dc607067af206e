// Self-test of the checkpoint layer without a GPU: writes a sedov-shaped checkpoint, reads it back and compares.
// usage: h5_selftest <file.h5>   (exit code 0 = round trip exact; 77 = libhdf5 not available)
#include <cmath>
#include <cstdio>
#include "h5_checkpoint.hpp"

int main(int argc, char** argv)
{
    if (argc != 2) return 2;
    if (! h5io::available()) return 77;
    const int nz = 7;
    std::vector<double> v(nz + 1), u(5 * nz);
    for (int i = 0; i <= nz; ++i) v[i] = std::pow(10.0, 0.1 * i);
    for (int i = 0; i < 5 * nz; ++i) u[i] = std::sin(i) * 1e-3;
    mara::config_t cfg = mara::config_t().item("outdir", "data").item("restart", "").item("nr", 256).item("tfinal", 1.5);
    h5io::schedule_t sched;
    sched.create_and_mark_as_due("write_checkpoint");
    sched.mark_as_completed("write_checkpoint");
    sched.advance("write_checkpoint", 2.5, 1.0);
    {
        auto file = h5io::Node::create_file(argv[1]);
        auto sol = file.require_group("solution");
        sol.write("time", 0.125);
        sol.write_rational("iteration", 42, 1);
        sol.write("vertices", v);
        sol.write_cells("conserved", {hsize_t(nz)}, 5, u.data());
        h5io::write_schedule(file.require_group("schedule"), sched);
        h5io::write_config(file.require_group("config"), cfg);
    }
    auto file = h5io::Node::open_file(argv[1]);
    auto sol = file.open_group("solution");
    int num = 0, den = 0;
    sol.read_rational("iteration", num, den);
    std::vector<hsize_t> shape;
    bool ok = sol.read_double("time") == 0.125 && num == 42 && den == 1 && sol.read_vector("vertices") == v
           && sol.read_cells("conserved", 5, shape) == u && shape.size() == 1 && shape[0] == hsize_t(nz);
    auto s2 = h5io::read_schedule(file.open_group("schedule"));
    ok = ok && s2.at("write_checkpoint").num_times_performed == 1 && s2.at("write_checkpoint").last_performed == 1.0 && ! s2.is_due("write_checkpoint");
    mara::config_t c2 = mara::config_t().item("outdir", "x").item("restart", "y").item("nr", 1).item("tfinal", 0.0).item("not_stored", 3);
    h5io::read_config_into(file.open_group("config"), c2);
    ok = ok && c2.get_string("outdir") == "data" && c2.get_string("restart") == "" && c2.get_int("nr") == 256 && c2.get_double("tfinal") == 1.5 && c2.get_int("not_stored") == 3;
    std::printf(ok ? "round trip ok\n" : "round trip FAILED\n");
    return ok ? 0 : 1;
}
