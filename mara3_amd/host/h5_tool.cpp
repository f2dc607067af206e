// h5_tool: the checkpoint layer of the compiled hosts (h5_checkpoint.hpp) behind a command line, without a GPU - the product-side twin of
// oracle/ref_drivers/h5_ref.cpp, which does the same with the reference's own writers and readers. Both speak one text "spec" of a
// checkpoint (doubles as C hex floats), so that the tests can ask, in both directions and bit for bit:
//     spec -> this writer -> file  ==  spec -> reference writer -> file        (h5dump of both, types and data)
//     this writer -> file -> reference reader -> spec'  ==  spec                (the reference's readers accept these files)
//     reference writer -> file -> this reader -> spec'' ==  spec               (restart=<reference-written file> works)
// The calls are the ones the hosts make: `sedov` subprog_sedov.cpp (write :176-186, read :108-119), `cloud` subprog_cloud.cpp (:227-235,
// :198-210), `binary` subprog_binary.cpp for the tree datasets and the orbital-element compounds.
//
//     kind sedov|cloud
//     time <hex>
//     iteration <num> <den>
//     array <name> <n> <hex> ...
//     conserved <rank> <n0> <n1> <hex> ...
//     task <name> <num_times_performed> <last_performed hex>
//     config <key> i <int> | d <hex> | s <length> <raw characters>
//
// usage: h5_tool write <spec> <out.h5> | read <kind> <in.h5> <spec> | tree_write <out.h5> | tree_read <in.h5> <out.txt>
//        | elements_write <out.h5> | elements_read <in.h5> <out.txt>          (exit code 77: libhdf5 not available)
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include "h5_checkpoint.hpp"
#include "mara_hip.h"

namespace {

struct spec_t
{
    std::string kind;
    double time = 0.0;
    int num = 0, den = 1;
    std::vector<std::pair<std::string, std::vector<double>>> arrays;
    int rank = 1;
    std::size_t n0 = 0, n1 = 1;
    std::vector<double> conserved;
    h5io::schedule_t schedule;
    mara::config_t config;
};

double from_hex(const std::string& s) { return std::strtod(s.c_str(), nullptr); }
std::string to_hex(double x) { char b[64]; std::snprintf(b, sizeof b, "%a", x); return b; }

spec_t load_spec(const char* path)
{
    std::ifstream in(path);
    if (! in) throw std::runtime_error(std::string("cannot open ") + path);
    spec_t spec;
    std::string line;
    while (std::getline(in, line))
    {
        std::istringstream ss(line);
        std::string word, h;
        if (! (ss >> word)) continue;
        if (word == "kind") ss >> spec.kind;
        else if (word == "time") { ss >> h; spec.time = from_hex(h); }
        else if (word == "iteration") ss >> spec.num >> spec.den;
        else if (word == "array")
        {
            std::string name; std::size_t n;
            ss >> name >> n;
            std::vector<double> v(n);
            for (auto& x : v) { ss >> h; x = from_hex(h); }
            spec.arrays.emplace_back(name, v);
        }
        else if (word == "conserved")
        {
            ss >> spec.rank >> spec.n0 >> spec.n1;
            spec.conserved.resize(spec.n0 * spec.n1 * 5);
            for (auto& x : spec.conserved) { ss >> h; x = from_hex(h); }
        }
        else if (word == "task")
        {
            h5io::schedule_t::task_t task;
            ss >> task.name >> task.num_times_performed >> h;
            task.last_performed = from_hex(h);
            spec.schedule.tasks[task.name] = task;
        }
        else if (word == "config")
        {
            std::string key, type;
            ss >> key >> type;
            if (type == "i") { int v; ss >> v; spec.config.item(key, v); }
            else if (type == "d") { ss >> h; spec.config.item(key, from_hex(h)); }
            else
            {
                std::size_t len; ss >> len;
                ss.get();
                std::string v(len, '\0');
                ss.read(v.data(), std::streamsize(len));
                spec.config.item(key, v);
            }
        }
        else throw std::runtime_error("spec: unknown line " + word);
    }
    return spec;
}

void dump_spec(const spec_t& spec, const char* path)
{
    std::ofstream out(path);
    out << "kind " << spec.kind << "\n" << "time " << to_hex(spec.time) << "\n" << "iteration " << spec.num << " " << spec.den << "\n";
    for (const auto& a : spec.arrays)
    {
        out << "array " << a.first << " " << a.second.size();
        for (double x : a.second) out << " " << to_hex(x);
        out << "\n";
    }
    out << "conserved " << spec.rank << " " << spec.n0 << " " << spec.n1;
    for (double x : spec.conserved) out << " " << to_hex(x);
    out << "\n";
    for (const auto& t : spec.schedule.tasks) out << "task " << t.first << " " << t.second.num_times_performed << " " << to_hex(t.second.last_performed) << "\n";
    for (const auto& kv : spec.config.items())
    {
        out << "config " << kv.first << " ";
        switch (kv.second.index())
        {
            case 0: out << "i " << std::get<int>(kv.second); break;
            case 1: out << "d " << to_hex(std::get<double>(kv.second)); break;
            case 2: out << "s " << std::get<std::string>(kv.second).size() << " " << std::get<std::string>(kv.second); break;
        }
        out << "\n";
    }
}

// the hosts' write_checkpoint, for either sub-program
void write_checkpoint(const spec_t& spec, const char* fname)
{
    auto file = h5io::Node::create_file(fname);
    auto sol = file.require_group("solution");
    sol.write("time", spec.time);
    sol.write_rational("iteration", spec.num, spec.den);
    for (const auto& a : spec.arrays) sol.write(a.first, a.second);
    if (spec.kind == "sedov") sol.write_cells("conserved", {hsize_t(spec.n0)}, 5, spec.conserved.data());
    else sol.write_cells("conserved", {hsize_t(spec.n0), hsize_t(spec.n1)}, 5, spec.conserved.data());
    h5io::write_schedule(file.require_group("schedule"), spec.schedule);
    h5io::write_config(file.require_group("config"), spec.config);
}

// the hosts' restart path; the configuration is read item by item with the type the file gives it (the hosts type it by their template)
spec_t read_checkpoint(const std::string& kind, const char* fname)
{
    spec_t spec;
    spec.kind = kind;
    auto file = h5io::Node::open_file(fname);
    auto sol = file.open_group("solution");
    sol.read_rational("iteration", spec.num, spec.den);
    spec.time = sol.read_double("time");
    for (const char* name : {"vertices", "radial_vertices", "polar_vertices"})
        if (sol.has(name)) spec.arrays.emplace_back(name, sol.read_vector(name));
    std::vector<hsize_t> shape;
    spec.conserved = sol.read_cells("conserved", 5, shape);
    spec.rank = int(shape.size());
    spec.n0 = shape.at(0);
    spec.n1 = shape.size() > 1 ? shape[1] : 1;
    spec.schedule = h5io::read_schedule(file.open_group("schedule"));
    auto cfg = file.open_group("config");
    for (const auto& name : cfg.names())
    {
        switch (cfg.class_of(name))
        {
            case H5T_INTEGER: spec.config.item(name, cfg.read_int(name)); break;
            case H5T_FLOAT:   spec.config.item(name, cfg.read_double(name)); break;
            case H5T_STRING:  spec.config.item(name, cfg.read_string(name)); break;
            default: throw std::runtime_error("config item of unknown class: " + name);
        }
    }
    return spec;
}

// the graded quadtree of oracle/ref_drivers/h5_ref.cpp (the chain of last children refined down to level 4), as the list of leaves that
// `mara_hip binary` holds (mh_tree_block: level, i, j), in the same cell values
struct leaf_t { int level, i, j; };

void collect(std::vector<leaf_t>& out, int level, int i, int j)
{
    const int last = (1 << level) - 1;
    const bool refine = level < 1 || (level < 4 && i == last && j == last);
    if (! refine) { out.push_back({level, i, j}); return; }
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) collect(out, level + 1, 2 * i + a, 2 * j + b);
}

double tree_cell_value(const leaf_t& l, int i, int j, int q)
{
    return double(l.level) * 1000.0 + double(l.i) * 100.0 + double(l.j) * 10.0 + double(i) * 0.25 + double(j) * 0.0625 + double(q) * 1e-3;
}

void tree_write(const char* fname)
{
    std::vector<leaf_t> leaves;
    collect(leaves, 0, 0, 0);
    auto file = h5io::Node::create_file(fname);
    auto g = file.require_group("conserved_u");
    for (const auto& l : leaves)
    {
        std::vector<double> cells(4 * 4 * 3);
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 3; ++q) cells[(i * 4 + j) * 3 + q] = tree_cell_value(l, i, j, q);
        g.write_cells(h5io::format_tree_index(l.level, l.i, l.j), {4, 4}, 3, cells.data());          // subprog_binary.cpp, write_checkpoint
    }
}

void tree_read(const char* fname, const char* out_name)
{
    auto file = h5io::Node::open_file(fname);
    auto g = file.open_group("conserved_u");
    std::vector<leaf_t> leaves;
    collect(leaves, 0, 0, 0);
    std::ofstream out(out_name);
    if (g.names().size() != leaves.size()) throw std::runtime_error("tree_read: the file holds another number of blocks than the tree has leaves");
    for (const auto& l : leaves)             // the restart path asks for each leaf of ITS tree by name
    {
        std::vector<hsize_t> shape;
        const auto cells = g.read_cells(h5io::format_tree_index(l.level, l.i, l.j), 3, shape);
        out << h5io::format_tree_index(l.level, l.i, l.j) << " " << shape.at(0) << " " << shape.at(1);
        for (double x : cells) out << " " << to_hex(x);
        out << "\n";
    }
}

struct element_types_t       // as record_types_t of subprog_binary.cpp
{
    h5io::Compound elements{sizeof(mh_orbital_elements)};
    h5io::Compound full{sizeof(mh_full_orbital_elements)};
    element_types_t()
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
    }
};

void elements_write(const char* fname)
{
    mh_full_orbital_elements E = {};
    E.pomega = 0.125; E.tau = -0.5; E.cm_position_x = 1e-3; E.cm_position_y = -2e-3; E.cm_velocity_x = 3e-4; E.cm_velocity_y = -4e-4;
    E.elements.separation = 1.0; E.elements.total_mass = 1.0; E.elements.mass_ratio = 0.75; E.elements.eccentricity = 0.1;
    element_types_t types;
    auto file = h5io::Node::create_file(fname);
    auto sol = file.require_group("solution");
    sol.write_record("orbital_elements", types.full, &E);
    sol.write_record("orbital_elements_kepler", types.elements, &E.elements);
}

void elements_read(const char* fname, const char* out_name)
{
    element_types_t types;
    auto file = h5io::Node::open_file(fname);
    auto sol = file.open_group("solution");
    mh_full_orbital_elements E = {};
    mh_orbital_elements K = {};
    sol.read_record("orbital_elements", types.full, &E);
    sol.read_record("orbital_elements_kepler", types.elements, &K);
    std::ofstream out(out_name);
    for (double x : {E.pomega, E.tau, E.cm_position_x, E.cm_position_y, E.cm_velocity_x, E.cm_velocity_y, E.elements.separation,
                     E.elements.total_mass, E.elements.mass_ratio, E.elements.eccentricity, K.separation, K.total_mass, K.mass_ratio, K.eccentricity})
        out << to_hex(x) << "\n";
}

} // namespace

int main(int argc, const char* argv[])
{
    const std::string mode = argc > 1 ? argv[1] : "";
    try
    {
        if (! h5io::available()) return 77;
        if (mode == "write" && argc == 4) write_checkpoint(load_spec(argv[2]), argv[3]);
        else if (mode == "read" && argc == 5) dump_spec(read_checkpoint(argv[2], argv[3]), argv[4]);
        else if (mode == "tree_write" && argc == 3) tree_write(argv[2]);
        else if (mode == "tree_read" && argc == 4) tree_read(argv[2], argv[3]);
        else if (mode == "elements_write" && argc == 3) elements_write(argv[2]);
        else if (mode == "elements_read" && argc == 4) elements_read(argv[2], argv[3]);
        else
        {
            std::fprintf(stderr, "usage: h5_tool write <spec> <out.h5> | read <kind> <in.h5> <spec> | tree_write <out.h5> | tree_read <in.h5> <out.txt> | elements_write <out.h5> | elements_read <in.h5> <out.txt>\n");
            return 2;
        }
    }
    catch (const std::exception& e)
    {
        std::fprintf(stderr, "h5_tool: %s\n", e.what());
        return 1;
    }
    return 0;
}
