/*
 * TEST INFRASTRUCTURE — reference-side oracle driver for the HDF5 file format.
 *
 * It calls the reference's OWN writers and readers, from its own headers where they lie (-I/root/reference/src):
 *   mara::write / mara::read, write_schedule / read_schedule, write_config / read_config      app_serialize.hpp:69-162
 *   the hdf5_type_info specialisations that decide types and dataspaces                       app_serialize.hpp:195-347
 *   mara::write_tree / read_tree, format_tree_index / read_tree_index                         app_serialize_tree.hpp:73-177
 *   h5::File / Group / Datatype::compound                                                     core_hdf5.hpp
 * in the sequences the sub-programs use: `sedov` write_solution / read_solution / write_checkpoint (subprog_sedov.cpp:329-346,
 * 486-495) and `cloud` (subprog_cloud.cpp:590-609,758-767). The compound types of `binary` (orbital elements) are specialised upstream in
 * subprog_binary_io.cpp:44-92, a translation unit that needs the generated app_compile_opts.hpp and is therefore out of reach: mode
 * `elements_*` below restates those two specialisations (member lists copied by NAME onto the reference's own structs of
 * model_two_body.hpp, through the reference's own h5::Datatype::compound and h5_compound_type_member) - a pin of the compound
 * machinery and of the struct layout, not of that file.
 *
 * A state travels as a small text "spec" (doubles as C hex floats, so every bit survives):
 *     kind sedov|cloud
 *     time <hex>
 *     iteration <num> <den>
 *     array <name> <n> <hex> ...                    1-D arrays of doubles (vertices), in the order they are written
 *     conserved <rank> <n0> <n1> <hex> ...           n0 * n1 cells of 5 doubles (rank 1: n1 = 1)
 *     task <name> <num_times_performed> <last_performed hex>
 *     config <key> i <int> | d <hex> | s <length> <raw characters>
 *
 * usage: h5_ref write <spec> <out.h5>          the reference writes the checkpoint described by <spec>
 *        h5_ref read <kind> <in.h5> <spec>     the reference reads a checkpoint; what it got goes to <spec>
 *        h5_ref tree_write <out.h5> | tree_read <in.h5> <out.txt>          a small graded quadtree of [4][4] blocks of double[3]
 *        h5_ref elements_write <out.h5> | elements_read <in.h5> <out.txt>  the two orbital-element compounds
 */
#include <cstring>
#include <algorithm>
#include <numeric>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include "app_serialize.hpp"
#include "app_serialize_tree.hpp"
#include "core_tree.hpp"
#include "core_sequence.hpp"
#include "model_two_body.hpp"
#include "physics_srhd.hpp"
#include "physics_euler.hpp"

using conserved_t = mara::srhd::conserved_t;      // == mara::euler::conserved_t: arithmetic_sequence_t<unit_mass<double>, 5>
static_assert(std::is_same<mara::srhd::conserved_t, mara::euler::conserved_t>::value, "one cell type for both hydro systems");


//=============================================================================
struct spec_t
{
    std::string kind;
    double time = 0.0;
    int num = 0, den = 1;
    std::vector<std::pair<std::string, std::vector<double>>> arrays;
    int rank = 1;
    std::size_t n0 = 0, n1 = 1;
    std::vector<double> conserved;
    mara::schedule_t schedule;
    mara::config_parameter_map_t config;
};

static double hex_to_double(const std::string& s) { return std::strtod(s.data(), nullptr); }
static std::string double_to_hex(double x) { char b[64]; std::snprintf(b, sizeof b, "%a", x); return b; }

static spec_t load_spec(const char* path)
{
    auto in = std::ifstream(path);
    if (! in) throw std::runtime_error(std::string("cannot open ") + path);
    auto spec = spec_t();
    auto line = std::string();
    while (std::getline(in, line))
    {
        auto ss = std::istringstream(line);
        auto word = std::string();
        if (! (ss >> word)) continue;
        if (word == "kind") ss >> spec.kind;
        else if (word == "time") { std::string h; ss >> h; spec.time = hex_to_double(h); }
        else if (word == "iteration") ss >> spec.num >> spec.den;
        else if (word == "array")
        {
            std::string name, h; std::size_t n;
            ss >> name >> n;
            auto v = std::vector<double>(n);
            for (auto& x : v) { ss >> h; x = hex_to_double(h); }
            spec.arrays.emplace_back(name, v);
        }
        else if (word == "conserved")
        {
            std::string h;
            ss >> spec.rank >> spec.n0 >> spec.n1;
            spec.conserved.resize(spec.n0 * spec.n1 * 5);
            for (auto& x : spec.conserved) { ss >> h; x = hex_to_double(h); }
        }
        else if (word == "task")
        {
            auto task = mara::schedule_t::task_t();
            std::string h;
            ss >> task.name >> task.num_times_performed >> h;
            task.last_performed = hex_to_double(h);
            spec.schedule.insert(task);
        }
        else if (word == "config")
        {
            std::string key, type;
            ss >> key >> type;
            if (type == "i") { int v; ss >> v; spec.config[key] = v; }
            else if (type == "d") { std::string h; ss >> h; spec.config[key] = hex_to_double(h); }
            else
            {
                std::size_t len; ss >> len;
                ss.get();                                        // the one blank after the length
                auto v = std::string(len, '\0');
                ss.read(v.data(), len);
                spec.config[key] = v;
            }
        }
        else throw std::runtime_error("spec: unknown line " + word);
    }
    return spec;
}

static void dump_spec(const spec_t& spec, const char* path)
{
    auto out = std::ofstream(path);
    out << "kind " << spec.kind << "\n";
    out << "time " << double_to_hex(spec.time) << "\n";
    out << "iteration " << spec.num << " " << spec.den << "\n";
    for (const auto& a : spec.arrays)
    {
        out << "array " << a.first << " " << a.second.size();
        for (auto x : a.second) out << " " << double_to_hex(x);
        out << "\n";
    }
    out << "conserved " << spec.rank << " " << spec.n0 << " " << spec.n1;
    for (auto x : spec.conserved) out << " " << double_to_hex(x);
    out << "\n";
    for (const auto& t : spec.schedule)
        out << "task " << t.first << " " << t.second.num_times_performed << " " << double_to_hex(t.second.last_performed) << "\n";
    for (const auto& item : spec.config)
    {
        out << "config " << item.first << " ";
        switch (item.second.index())
        {
            case 0: out << "i " << std::get<0>(item.second); break;
            case 1: out << "d " << double_to_hex(std::get<1>(item.second)); break;
            case 2: out << "s " << std::get<2>(item.second).size() << " " << std::get<2>(item.second); break;
        }
        out << "\n";
    }
}

template<typename T>
static auto array_1d(const std::vector<double>& v)
{
    return nd::make_array([v] (auto i) { return T(v[i[0]]); }, nd::make_shape(v.size())) | nd::to_shared();
}

static auto cell_at(const std::vector<double>& u, std::size_t flat)
{
    auto c = conserved_t();
    for (std::size_t q = 0; q < 5; ++q) c[q] = mara::make_mass(u[flat * 5 + q]);
    return c;
}


//=============================================================================
// `sedov`: write_solution (subprog_sedov.cpp:329-335) inside write_checkpoint (:486-495), read_solution (:338-346)
static void write_sedov(const spec_t& spec, const char* fname)
{
    auto vertices  = array_1d<double>(spec.arrays.at(0).second);
    auto conserved = nd::make_array([&spec] (auto i) { return cell_at(spec.conserved, i[0]); }, nd::make_shape(spec.n0)) | nd::to_shared();
    auto iteration = mara::make_rational(spec.num, spec.den);
    auto file  = h5::File(fname, "w");
    auto group = file.require_group("solution");
    group.write("time", spec.time);
    group.write("iteration", iteration);
    group.write("vertices", vertices);
    group.write("conserved", conserved);
    mara::write_schedule(file.require_group("schedule"), spec.schedule);
    mara::write_config(file.require_group("config"), mara::config_t(spec.config, spec.config));
}

static spec_t read_sedov(const char* fname)
{
    auto spec  = spec_t();
    auto file  = h5::File(fname, "r");
    auto group = file.open_group("solution");
    auto iteration = mara::make_rational(0, 1);
    auto vertices  = nd::shared_array<double, 1>();
    auto conserved = nd::shared_array<conserved_t, 1>();
    group.read("time", spec.time);
    group.read("iteration", iteration);
    group.read("vertices", vertices);
    group.read("conserved", conserved);
    spec.kind = "sedov";
    spec.num = iteration.get_numerator();
    spec.den = iteration.get_denominator();
    spec.arrays.emplace_back("vertices", std::vector<double>(vertices.begin(), vertices.end()));
    spec.rank = 1; spec.n0 = conserved.shape(0); spec.n1 = 1;
    for (auto c : conserved) for (std::size_t q = 0; q < 5; ++q) spec.conserved.push_back(c[q].value);
    spec.schedule = mara::read_schedule(file.open_group("schedule"));
    spec.config   = mara::read_config(file.open_group("config"));
    return spec;
}


//=============================================================================
// `cloud`: write_solution (subprog_cloud.cpp:590-597) inside write_checkpoint (:758-767), read_solution (:599-609)
static void write_cloud(const spec_t& spec, const char* fname)
{
    auto radial_vertices = array_1d<mara::unit_length<double>>(spec.arrays.at(0).second);
    auto polar_vertices  = array_1d<double>(spec.arrays.at(1).second);
    auto n1 = spec.n1;
    auto conserved = nd::make_array([&spec, n1] (auto i) { return cell_at(spec.conserved, i[0] * n1 + i[1]); }, nd::make_shape(spec.n0, spec.n1)) | nd::to_shared();
    auto iteration = mara::make_rational(spec.num, spec.den);
    auto file  = h5::File(fname, "w");
    auto group = file.require_group("solution");
    group.write("time", spec.time);
    group.write("iteration", iteration);
    group.write("radial_vertices", radial_vertices);
    group.write("polar_vertices", polar_vertices);
    group.write("conserved", conserved);
    mara::write_schedule(file.require_group("schedule"), spec.schedule);
    mara::write_config(file.require_group("config"), mara::config_t(spec.config, spec.config));
}

static spec_t read_cloud(const char* fname)
{
    auto spec  = spec_t();
    auto file  = h5::File(fname, "r");
    auto group = file.open_group("solution");
    auto iteration = mara::make_rational(0, 1);
    auto radial_vertices = nd::shared_array<mara::unit_length<double>, 1>();
    auto polar_vertices  = nd::shared_array<double, 1>();
    auto conserved = nd::shared_array<conserved_t, 2>();
    group.read("time", spec.time);
    group.read("iteration", iteration);
    group.read("radial_vertices", radial_vertices);
    group.read("polar_vertices", polar_vertices);
    group.read("conserved", conserved);
    spec.kind = "cloud";
    spec.num = iteration.get_numerator();
    spec.den = iteration.get_denominator();
    auto rv = std::vector<double>();
    for (auto r : radial_vertices) rv.push_back(r.value);
    spec.arrays.emplace_back("radial_vertices", rv);
    spec.arrays.emplace_back("polar_vertices", std::vector<double>(polar_vertices.begin(), polar_vertices.end()));
    spec.rank = 2; spec.n0 = conserved.shape(0); spec.n1 = conserved.shape(1);
    for (auto c : conserved) for (std::size_t q = 0; q < 5; ++q) spec.conserved.push_back(c[q].value);
    spec.schedule = mara::read_schedule(file.open_group("schedule"));
    spec.config   = mara::read_config(file.open_group("config"));
    return spec;
}


//=============================================================================
// A graded quadtree (the chain of last children refined down to level 4, so that names with zero-padded coordinates such as "4:14-15"
// occur) of [4][4] blocks whose cells are double[3]: the on-disk form of `binary`'s
// conserved trees (H5T_ARRAY{3 x f64} per cell, subprog_binary_io.cpp:10-20), written and read by write_tree / read_tree.
using vec3 = mara::arithmetic_sequence_t<double, 3>;
using block_tree_t = mara::arithmetic_binary_tree_t<nd::shared_array<vec3, 2>, 2>;

static double tree_cell_value(const mara::tree_index_t<2>& index, std::size_t i, std::size_t j, std::size_t q)
{
    return double(index.level) * 1000.0 + double(index.coordinates[0]) * 100.0 + double(index.coordinates[1]) * 10.0 + double(i) * 0.25 + double(j) * 0.0625 + double(q) * 1e-3;
}

static block_tree_t make_block_tree()
{
    auto refine = [] (auto index)
    {
        auto last = std::size_t((1 << index.level) - 1);
        return index.level < 1 || (index.level < 4 && index.coordinates[0] == last && index.coordinates[1] == last);
    };
    auto indexes = mara::tree_of<2>(mara::tree_index_t<2>());
    for (int pass = 0; pass < 4; ++pass)           // bifurcate_if visits the leaves once per call (as create_vertex_quadtree loops over depth)
        indexes = std::move(indexes).bifurcate_if(refine, [] (auto index) { return index.child_indexes(); });
    return indexes.map([] (auto index)
    {
        return nd::make_array([index] (auto ij)
        {
            auto c = vec3();
            for (std::size_t q = 0; q < 3; ++q) c[q] = tree_cell_value(index, ij[0], ij[1], q);
            return c;
        }, nd::make_shape(4, 4)) | nd::to_shared();
    });
}

static void tree_write(const char* fname)
{
    auto file = h5::File(fname, "w");
    auto root = file.open_group("/");
    mara::write(root, "conserved_u", make_block_tree());
}

static void tree_read(const char* fname, const char* out_name)
{
    auto file = h5::File(fname, "r");
    auto root = file.open_group("/");
    auto tree = block_tree_t();
    mara::read(root, "conserved_u", tree);
    auto out = std::ofstream(out_name);
    tree.indexes().pair(tree).sink([&out] (auto&& index_and_block)
    {
        auto [index, block] = index_and_block;
        out << mara::format_tree_index(index) << " " << block.shape(0) << " " << block.shape(1);
        for (auto c : block) for (std::size_t q = 0; q < 3; ++q) out << " " << double_to_hex(c[q]);
        out << "\n";
    });
}


//=============================================================================
// The orbital-element compounds, member lists as in subprog_binary_io.cpp:44-92 (restated: that file cannot be included)
template<>
struct h5::hdf5_type_info<mara::orbital_elements_t>
{
    using native_type = mara::orbital_elements_t;
    static auto make_datatype_for(const native_type&)
    {
        return h5::Datatype::compound<native_type>({
            h5_compound_type_member(native_type, separation),
            h5_compound_type_member(native_type, total_mass),
            h5_compound_type_member(native_type, mass_ratio),
            h5_compound_type_member(native_type, eccentricity),
        });
    }
    static auto make_dataspace_for(const native_type&) { return Dataspace::scalar(); }
    static auto convert_to_writable(const native_type& value) { return value; }
    static auto prepare(const Datatype&, const Dataspace&) { return native_type(); }
    static auto finalize(native_type&& value) { return std::move(value); }
    static auto get_address(const native_type& value) { return &value; }
    static auto get_address(native_type& value) { return &value; }
};

template<>
struct h5::hdf5_type_info<mara::full_orbital_elements_t>
{
    using native_type = mara::full_orbital_elements_t;
    static auto make_datatype_for(const native_type&)
    {
        return h5::Datatype::compound<native_type>({
            h5_compound_type_member(native_type, pomega),
            h5_compound_type_member(native_type, tau),
            h5_compound_type_member(native_type, cm_position_x),
            h5_compound_type_member(native_type, cm_position_y),
            h5_compound_type_member(native_type, cm_velocity_x),
            h5_compound_type_member(native_type, cm_velocity_y),
            h5_compound_type_member(native_type, elements),
        });
    }
    static auto make_dataspace_for(const native_type&) { return Dataspace::scalar(); }
    static auto convert_to_writable(const native_type& value) { return value; }
    static auto prepare(const Datatype&, const Dataspace&) { return native_type(); }
    static auto finalize(native_type&& value) { return std::move(value); }
    static auto get_address(const native_type& value) { return &value; }
    static auto get_address(native_type& value) { return &value; }
};

static mara::full_orbital_elements_t sample_elements()
{
    auto E = mara::full_orbital_elements_t();
    E.pomega = 0.125; E.tau = -0.5; E.cm_position_x = 1e-3; E.cm_position_y = -2e-3; E.cm_velocity_x = 3e-4; E.cm_velocity_y = -4e-4;
    E.elements.separation = 1.0; E.elements.total_mass = 1.0; E.elements.mass_ratio = 0.75; E.elements.eccentricity = 0.1;
    return E;
}

static void elements_write(const char* fname)
{
    auto file = h5::File(fname, "w");
    auto group = file.require_group("solution");
    group.write("orbital_elements", sample_elements());
    group.write("orbital_elements_kepler", sample_elements().elements);
}

static void elements_read(const char* fname, const char* out_name)
{
    auto file = h5::File(fname, "r");
    auto group = file.open_group("solution");
    auto E = group.read<mara::full_orbital_elements_t>("orbital_elements");
    auto K = group.read<mara::orbital_elements_t>("orbital_elements_kepler");
    auto out = std::ofstream(out_name);
    for (double x : {E.pomega, E.tau, E.cm_position_x, E.cm_position_y, E.cm_velocity_x, E.cm_velocity_y, E.elements.separation,
                     E.elements.total_mass, E.elements.mass_ratio, E.elements.eccentricity, K.separation, K.total_mass, K.mass_ratio, K.eccentricity})
        out << double_to_hex(x) << "\n";
}


//=============================================================================
int main(int argc, const char* argv[])
{
    auto mode = std::string(argc > 1 ? argv[1] : "");
    try
    {
        if (mode == "write" && argc == 4)
        {
            auto spec = load_spec(argv[2]);
            if (spec.kind == "sedov") write_sedov(spec, argv[3]);
            else if (spec.kind == "cloud") write_cloud(spec, argv[3]);
            else throw std::runtime_error("unknown kind " + spec.kind);
        }
        else if (mode == "read" && argc == 5)
        {
            auto kind = std::string(argv[2]);
            dump_spec(kind == "sedov" ? read_sedov(argv[3]) : read_cloud(argv[3]), argv[4]);
        }
        else if (mode == "tree_write" && argc == 3) tree_write(argv[2]);
        else if (mode == "tree_read" && argc == 4) tree_read(argv[2], argv[3]);
        else if (mode == "elements_write" && argc == 3) elements_write(argv[2]);
        else if (mode == "elements_read" && argc == 4) elements_read(argv[2], argv[3]);
        else
        {
            std::fprintf(stderr, "usage: h5_ref write <spec> <out.h5> | read <kind> <in.h5> <spec> | tree_write <out.h5> | tree_read <in.h5> <out.txt> | elements_write <out.h5> | elements_read <in.h5> <out.txt>\n");
            return 2;
        }
    }
    catch (const std::exception& e)
    {
        std::fprintf(stderr, "h5_ref: %s\n", e.what());
        return 1;
    }
    return 0;
}
