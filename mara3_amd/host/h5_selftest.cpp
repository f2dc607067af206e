// Self-test of the checkpoint layer without a GPU: writes a sedov-shaped checkpoint, reads it back and compares.
// usage: h5_selftest <file.h5>   (exit code 0 = round trip exact; 77 = libhdf5 not available)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstddef>
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
    // tree dataset names: the reference's own known answers (src/app_test.cpp:377-384, two of the three coordinates)
    ok = ok && h5io::format_tree_index(0, 0, 0) == "0:0-0" && h5io::format_tree_index(3, 5, 6) == "3:5-6"
            && h5io::format_tree_index(5, 1, 16) == "5:01-16" && h5io::format_tree_index(8, 1, 2) == "8:001-002";
    // compound records (binary's orbital elements / time series) and extendible columns (sedov's time_series.h5)
    struct inner_t { double a, b; };
    struct record_t { double x; double v[2]; inner_t in; };
    {
        h5io::Compound inner(sizeof(inner_t)), rec(sizeof(record_t));
        inner.insert_double("a", offsetof(inner_t, a));
        inner.insert_double("b", offsetof(inner_t, b));
        rec.insert_double("x", offsetof(record_t, x));
        rec.insert_array("v", offsetof(record_t, v), 2);
        rec.insert("in", offsetof(record_t, in), inner);
        const record_t rows[3] = {{1, {2, 3}, {4, 5}}, {6, {7, 8}, {9, 10}}, {11, {12, 13}, {14, 15}}};
        const std::string path2 = std::string(argv[1]) + ".records.h5";
        {
            auto f2 = h5io::Node::create_file(path2);
            f2.write_records("rows", rec, 3, rows);
            f2.write_records("none", rec, 0, nullptr);
            f2.write_record("one", rec, &rows[1]);
            f2.create_unlimited("column", 1000);
        }
        {
            auto f2 = h5io::Node::open_file_rw(path2);
            f2.append("column", 0, 1.5);
            f2.append("column", 1, 2.5);
            f2.append("column", 4, 5.5);          // rows 2, 3 stay at the fill value
        }
        auto f2 = h5io::Node::open_file(path2);
        record_t back[3] = {}, one = {};
        f2.read_records("rows", rec, back);
        f2.read_record("one", rec, &one);
        ok = ok && f2.count_of("rows") == 3 && f2.count_of("none") == 0 && std::memcmp(back, rows, sizeof rows) == 0 && std::memcmp(&one, &rows[1], sizeof one) == 0;
        ok = ok && f2.read_vector("column") == std::vector<double>({1.5, 2.5, 0.0, 0.0, 5.5});
    }
    std::printf(ok ? "round trip ok\n" : "round trip FAILED\n");
    return ok ? 0 : 1;
}
