// HDF5 checkpoint / restart files of the compiled hosts, in the reference's layout:
//
//   chkpt.NNNN.h5
//     /solution/time                  f64 scalar
//     /solution/iteration             int[2] scalar        mara::rational_number_t   (app_serialize.hpp:274-284)
//     /solution/vertices | radial_vertices, polar_vertices   f64 [n]
//     /solution/conserved             [nz] or [nr][nq] of H5T_ARRAY{5 x f64}          (arithmetic_sequence_t, app_serialize.hpp:240-252)
//     /schedule/<task>/{name (C string), num_times_performed (int), last_performed (f64)}   (write_schedule :61-70)
//     /config/<item>                  int | f64 | fixed-length C string (size max(1, len)), scalar   (write_config :90-96, core_hdf5.hpp:474-477)
//
// written by subprog_sedov.cpp:329-335,486-495 and subprog_cloud.cpp:590-597,758-767 upstream. `binary`'s files (checkpoint with one
// dataset per tree block, compound orbital elements, the time series; diagnostics) are laid out in subprog_binary.cpp from the
// same pieces: scalar H5T_ARRAY datasets, 2-D datasets, and compound types (Compound below = h5::Datatype::compound, core_hdf5.hpp:197-210). The reference wraps the HDF5 C
// API in core_hdf5.hpp; here the C API is called directly. libhdf5 is bound at run time (dlopen), like RCCL in slab.hip: the
// host executable has no link-time dependency on it and runs without it as long as no checkpoint is requested.
//
// Parity status of this file format: PINNED in both directions for sedov / cloud checkpoints, schedule and config groups, tree datasets and
// the orbital-element compounds: oracle/ref_drivers/h5_ref.cpp calls the reference's own writers and readers (app_serialize.hpp,
// app_serialize_tree.hpp, core_hdf5.hpp compile against the image's libhdf5), h5_tool.cpp does the same through this header, and the two
// produce the same file and read each other's (tests/test_h5_format_cpu.py, tests/test_gpu_host_subprograms.py). Out of reach: the compound
// specialisations of subprog_binary_io.cpp (that translation unit needs the generated app_compile_opts.hpp).
#pragma once
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "app_config.hpp"
#if __has_include(<hdf5.h>)
#include <hdf5.h>
#define MH_HOST_HAVE_HDF5 1
#else
#define MH_HOST_HAVE_HDF5 0
#endif

#if ! MH_HOST_HAVE_HDF5
using hsize_t = unsigned long long;
#endif

namespace h5io {

#if ! MH_HOST_HAVE_HDF5
// Built where hdf5.h is absent: same interface, no checkpoints (available() is false, restart= reports the reason).
inline bool available() { std::printf("this build has no HDF5 support (hdf5.h was absent at compile time); no checkpoints will be written\n"); return false; }
struct Node
{
    [[noreturn]] static void none() { throw std::runtime_error("this build has no HDF5 support (hdf5.h was absent at compile time)"); }
    static Node create_file(const std::string&) { none(); }
    static Node open_file(const std::string&) { none(); }
    bool has(const std::string&) const { none(); }
    Node require_group(const std::string&) const { none(); }
    Node open_group(const std::string&) const { none(); }
    std::vector<std::string> names() const { none(); }
    template<typename T> void write(const std::string&, const T&) const { none(); }
    void write_rational(const std::string&, int, int) const { none(); }
    void write_cells(const std::string&, const std::vector<hsize_t>&, int, const double*) const { none(); }
    double read_double(const std::string&) const { none(); }
    int read_int(const std::string&) const { none(); }
    std::string read_string(const std::string&) const { none(); }
    void read_rational(const std::string&, int&, int&) const { none(); }
    std::vector<double> read_vector(const std::string&) const { none(); }
    std::vector<double> read_cells(const std::string&, int, std::vector<hsize_t>&) const { none(); }
    void write_array(const std::string&, const double*, int) const { none(); }
    void read_array(const std::string&, double*, int) const { none(); }
    void write_grid(const std::string&, hsize_t, hsize_t, const double*) const { none(); }
    template<typename C> void write_record(const std::string&, const C&, const void*) const { none(); }
    template<typename C> void write_records(const std::string&, const C&, hsize_t, const void*) const { none(); }
    template<typename C> void read_record(const std::string&, const C&, void*) const { none(); }
    hsize_t count_of(const std::string&) const { none(); }
    template<typename C> void read_records(const std::string&, const C&, void*) const { none(); }
    static Node open_file_rw(const std::string&) { none(); }
    void create_unlimited(const std::string&, hsize_t) const { none(); }
    void append(const std::string&, hsize_t, double) const { none(); }
};
struct Compound
{
    explicit Compound(std::size_t) {}
    void insert_double(const char*, std::size_t) {}
    void insert_array(const char*, std::size_t, int) {}
    void insert(const char*, std::size_t, const Compound&) {}
};
#else

// ---- run-time binding of the HDF5 C library ----------------------------------------------------------------------------
struct Lib
{
    void* handle = nullptr;
#define H5IO_FN(name) decltype(&::name) name = nullptr
    H5IO_FN(H5open); H5IO_FN(H5Fcreate); H5IO_FN(H5Fopen); H5IO_FN(H5Fclose); H5IO_FN(H5Gcreate2); H5IO_FN(H5Gopen2); H5IO_FN(H5Gclose);
    H5IO_FN(H5Lexists); H5IO_FN(H5Screate); H5IO_FN(H5Screate_simple); H5IO_FN(H5Sclose); H5IO_FN(H5Sget_simple_extent_ndims);
    H5IO_FN(H5Sget_simple_extent_dims); H5IO_FN(H5Tcopy); H5IO_FN(H5Tset_size); H5IO_FN(H5Tarray_create2); H5IO_FN(H5Tclose);
    H5IO_FN(H5Tget_size); H5IO_FN(H5Tget_class); H5IO_FN(H5Dcreate2); H5IO_FN(H5Dopen2); H5IO_FN(H5Dwrite); H5IO_FN(H5Dread);
    H5IO_FN(H5Dget_space); H5IO_FN(H5Dget_type); H5IO_FN(H5Dclose); H5IO_FN(H5Literate); H5IO_FN(H5Tcreate); H5IO_FN(H5Tinsert);
    H5IO_FN(H5Sget_simple_extent_npoints); H5IO_FN(H5Pcreate); H5IO_FN(H5Pset_chunk); H5IO_FN(H5Pclose); H5IO_FN(H5Dset_extent);
    H5IO_FN(H5Sselect_hyperslab);
#undef H5IO_FN
    hid_t native_double = -1, native_int = -1, c_s1 = -1, dataset_create = -1;

    static Lib& get()
    {
        static Lib lib;
        if (! lib.handle)
        {
            for (const char* name : {"libhdf5.so.103", "libhdf5.so", "/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5.so"})
                if ((lib.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
            if (! lib.handle) throw std::runtime_error("HDF5 is not available: libhdf5.so could not be loaded (checkpoint / restart need it)");
#define H5IO_SYM(name) if (! (lib.name = reinterpret_cast<decltype(lib.name)>(dlsym(lib.handle, #name)))) throw std::runtime_error("libhdf5 lacks " #name)
            H5IO_SYM(H5open); H5IO_SYM(H5Fcreate); H5IO_SYM(H5Fopen); H5IO_SYM(H5Fclose); H5IO_SYM(H5Gcreate2); H5IO_SYM(H5Gopen2); H5IO_SYM(H5Gclose);
            H5IO_SYM(H5Lexists); H5IO_SYM(H5Screate); H5IO_SYM(H5Screate_simple); H5IO_SYM(H5Sclose); H5IO_SYM(H5Sget_simple_extent_ndims);
            H5IO_SYM(H5Sget_simple_extent_dims); H5IO_SYM(H5Tcopy); H5IO_SYM(H5Tset_size); H5IO_SYM(H5Tarray_create2); H5IO_SYM(H5Tclose);
            H5IO_SYM(H5Tget_size); H5IO_SYM(H5Tget_class); H5IO_SYM(H5Dcreate2); H5IO_SYM(H5Dopen2); H5IO_SYM(H5Dwrite); H5IO_SYM(H5Dread);
            H5IO_SYM(H5Dget_space); H5IO_SYM(H5Dget_type); H5IO_SYM(H5Dclose); H5IO_SYM(H5Literate); H5IO_SYM(H5Tcreate); H5IO_SYM(H5Tinsert);
            H5IO_SYM(H5Sget_simple_extent_npoints); H5IO_SYM(H5Pcreate); H5IO_SYM(H5Pset_chunk); H5IO_SYM(H5Pclose); H5IO_SYM(H5Dset_extent);
            H5IO_SYM(H5Sselect_hyperslab);
#undef H5IO_SYM
            lib.H5open();
            auto global = [&] (const char* sym) { auto p = static_cast<hid_t*>(dlsym(lib.handle, sym)); if (! p) throw std::runtime_error(std::string("libhdf5 lacks ") + sym); return *p; };
            lib.native_double = global("H5T_NATIVE_DOUBLE_g");
            lib.native_int = global("H5T_NATIVE_INT_g");
            lib.c_s1 = global("H5T_C_S1_g");
            lib.dataset_create = global("H5P_CLS_DATASET_CREATE_ID_g");
        }
        return lib;
    }
};

// false (with one note on stdout) where libhdf5 cannot be loaded: the hosts then run without writing checkpoints
inline bool available()
{
    try { Lib::get(); return true; }
    catch (const std::exception& e) { std::printf("%s; no checkpoints will be written\n", e.what()); return false; }
}

inline void check(hid_t id, const std::string& what) { if (id < 0) throw std::runtime_error("HDF5: " + what + " failed"); }

// h5::Datatype::compound (core_hdf5.hpp:197-210): a struct of doubles, arrays of doubles and nested compounds, members at the
// offsets of the in-memory struct
struct Compound
{
    hid_t id = -1;
    explicit Compound(std::size_t size) : id(Lib::get().H5Tcreate(H5T_COMPOUND, size)) { check(id, "create compound type"); }
    Compound(const Compound&) = delete;
    ~Compound() { if (id >= 0) Lib::get().H5Tclose(id); }
    void insert_double(const char* name, std::size_t offset) { Lib::get().H5Tinsert(id, name, offset, Lib::get().native_double); }
    void insert_array(const char* name, std::size_t offset, int n)
    {
        auto& L = Lib::get();
        const hsize_t q = n;
        hid_t t = L.H5Tarray_create2(L.native_double, 1, &q);
        L.H5Tinsert(id, name, offset, t);
        L.H5Tclose(t);
    }
    void insert(const char* name, std::size_t offset, const Compound& member) { Lib::get().H5Tinsert(id, name, offset, member.id); }
};

// a file or group handle
struct Node
{
    hid_t id = -1;
    bool is_file = false;
    Node() {}
    Node(hid_t i, bool f) : id(i), is_file(f) {}
    Node(const Node&) = delete;
    Node(Node&& o) : id(o.id), is_file(o.is_file) { o.id = -1; }
    ~Node() { if (id >= 0) { if (is_file) Lib::get().H5Fclose(id); else Lib::get().H5Gclose(id); } }

    static Node create_file(const std::string& path)
    {
        hid_t f = Lib::get().H5Fcreate(path.c_str(), 0x0002u /* H5F_ACC_TRUNC without the macro's H5open() */, H5P_DEFAULT, H5P_DEFAULT);
        check(f, "create " + path);
        return Node(f, true);
    }
    static Node open_file(const std::string& path)
    {
        hid_t f = Lib::get().H5Fopen(path.c_str(), 0x0000u /* H5F_ACC_RDONLY */, H5P_DEFAULT);
        check(f, "open " + path);
        return Node(f, true);
    }
    static Node open_file_rw(const std::string& path)
    {
        hid_t f = Lib::get().H5Fopen(path.c_str(), 0x0001u /* H5F_ACC_RDWR */, H5P_DEFAULT);
        check(f, "open " + path);
        return Node(f, true);
    }
    bool has(const std::string& name) const { return Lib::get().H5Lexists(id, name.c_str(), H5P_DEFAULT) > 0; }
    Node require_group(const std::string& name) const
    {
        auto& L = Lib::get();
        hid_t g = has(name) ? L.H5Gopen2(id, name.c_str(), H5P_DEFAULT) : L.H5Gcreate2(id, name.c_str(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        check(g, "group " + name);
        return Node(g, false);
    }
    Node open_group(const std::string& name) const
    {
        hid_t g = Lib::get().H5Gopen2(id, name.c_str(), H5P_DEFAULT);
        check(g, "open group " + name);
        return Node(g, false);
    }
    std::vector<std::string> names() const
    {
        std::vector<std::string> out;
        auto op = [] (hid_t, const char* name, const H5L_info_t*, void* data) -> herr_t { static_cast<std::vector<std::string>*>(data)->push_back(name); return 0; };
        Lib::get().H5Literate(id, H5_INDEX_NAME, H5_ITER_INC, nullptr, op, &out);
        return out;
    }

    // ---- writers
    void write_raw(const std::string& name, hid_t type, int rank, const hsize_t* dims, const void* data) const
    {
        auto& L = Lib::get();
        hid_t space = rank == 0 ? L.H5Screate(H5S_SCALAR) : L.H5Screate_simple(rank, dims, nullptr);
        hid_t ds = L.H5Dcreate2(id, name.c_str(), type, space, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        check(ds, "create dataset " + name);
        if (L.H5Dwrite(ds, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0) throw std::runtime_error("HDF5: write " + name + " failed");
        L.H5Dclose(ds);
        L.H5Sclose(space);
    }
    void write(const std::string& name, double v) const { write_raw(name, Lib::get().native_double, 0, nullptr, &v); }
    void write(const std::string& name, int v) const { write_raw(name, Lib::get().native_int, 0, nullptr, &v); }
    void write(const std::string& name, const std::string& v) const
    {
        auto& L = Lib::get();
        hid_t t = L.H5Tcopy(L.c_s1);
        L.H5Tset_size(t, std::max<std::size_t>(1, v.size()));          // core_hdf5.hpp:477
        std::string buf = v.empty() ? std::string(1, '\0') : v;
        write_raw(name, t, 0, nullptr, buf.data());
        L.H5Tclose(t);
    }
    void write_rational(const std::string& name, int num, int den) const
    {
        auto& L = Lib::get();
        const hsize_t two = 2;
        hid_t t = L.H5Tarray_create2(L.native_int, 1, &two);
        const int v[2] = {num, den};
        write_raw(name, t, 0, nullptr, v);
        L.H5Tclose(t);
    }
    void write(const std::string& name, const std::vector<double>& v) const
    {
        const hsize_t n = v.size();
        write_raw(name, Lib::get().native_double, 1, &n, v.data());
    }
    // cells[shape...][nq] as a dataset of shape `shape` whose element is an array of nq doubles
    void write_cells(const std::string& name, const std::vector<hsize_t>& shape, int nq, const double* data) const
    {
        auto& L = Lib::get();
        const hsize_t q = nq;
        hid_t t = L.H5Tarray_create2(L.native_double, 1, &q);
        write_raw(name, t, (int) shape.size(), shape.data(), data);
        L.H5Tclose(t);
    }

    // a scalar dataset whose element is an array of n doubles: mara::arithmetic_sequence_t (app_serialize.hpp:240-252)
    void write_array(const std::string& name, const double* v, int n) const
    {
        auto& L = Lib::get();
        const hsize_t q = n;
        hid_t t = L.H5Tarray_create2(L.native_double, 1, &q);
        write_raw(name, t, 0, nullptr, v);
        L.H5Tclose(t);
    }
    void write_grid(const std::string& name, hsize_t n0, hsize_t n1, const double* data) const
    {
        const hsize_t dims[2] = {n0, n1};
        write_raw(name, Lib::get().native_double, 2, dims, data);
    }
    void write_record(const std::string& name, const Compound& type, const void* data) const { write_raw(name, type.id, 0, nullptr, data); }
    // std::vector<record> (core_hdf5.hpp:487-497): a 1-D dataset, possibly of extent 0
    void write_records(const std::string& name, const Compound& type, hsize_t n, const void* data) const
    {
        static const char nothing[8] = {0};
        write_raw(name, type.id, 1, &n, n ? data : nothing);
    }

    // An empty, extendible 1-D f64 dataset (sedov's time_series.h5 columns: Dataspace::unlimited(0) with chunks of 1000,
    // src/subprog_sedov.cpp:605-612) and one value written at `index` after growing it (write_time_series :516-529)
    void create_unlimited(const std::string& name, hsize_t chunk) const
    {
        auto& L = Lib::get();
        const hsize_t zero = 0, unlimited = ~hsize_t(0) /* H5S_UNLIMITED */;
        hid_t space = L.H5Screate_simple(1, &zero, &unlimited);
        hid_t plist = L.H5Pcreate(L.dataset_create);
        L.H5Pset_chunk(plist, 1, &chunk);
        hid_t ds = L.H5Dcreate2(id, name.c_str(), L.native_double, space, H5P_DEFAULT, plist, H5P_DEFAULT);
        check(ds, "create dataset " + name);
        L.H5Dclose(ds);
        L.H5Pclose(plist);
        L.H5Sclose(space);
    }
    void append(const std::string& name, hsize_t index, double value) const
    {
        auto& L = Lib::get();
        hid_t ds = L.H5Dopen2(id, name.c_str(), H5P_DEFAULT);
        check(ds, "open dataset " + name);
        const hsize_t size = index + 1, one = 1;
        if (L.H5Dset_extent(ds, &size) < 0) throw std::runtime_error("HDF5: cannot extend " + name);
        hid_t fspace = L.H5Dget_space(ds);
        L.H5Sselect_hyperslab(fspace, H5S_SELECT_SET, &index, nullptr, &one, nullptr);
        hid_t mspace = L.H5Screate_simple(1, &one, nullptr);
        if (L.H5Dwrite(ds, L.native_double, mspace, fspace, H5P_DEFAULT, &value) < 0) throw std::runtime_error("HDF5: write " + name + " failed");
        L.H5Sclose(mspace);
        L.H5Sclose(fspace);
        L.H5Dclose(ds);
    }

    // ---- readers
    struct Dataset
    {
        hid_t ds = -1, space = -1, type = -1;
        ~Dataset() { auto& L = Lib::get(); if (type >= 0) L.H5Tclose(type); if (space >= 0) L.H5Sclose(space); if (ds >= 0) L.H5Dclose(ds); }
    };
    void open(const std::string& name, Dataset& d) const
    {
        auto& L = Lib::get();
        d.ds = L.H5Dopen2(id, name.c_str(), H5P_DEFAULT);
        check(d.ds, "open dataset " + name);
        d.space = L.H5Dget_space(d.ds);
        d.type = L.H5Dget_type(d.ds);
    }
    double read_double(const std::string& name) const
    {
        Dataset d; open(name, d);
        double v = 0;
        if (Lib::get().H5Dread(d.ds, Lib::get().native_double, H5S_ALL, H5S_ALL, H5P_DEFAULT, &v) < 0) throw std::runtime_error("HDF5: read " + name);
        return v;
    }
    int read_int(const std::string& name) const
    {
        Dataset d; open(name, d);
        int v = 0;
        if (Lib::get().H5Dread(d.ds, Lib::get().native_int, H5S_ALL, H5S_ALL, H5P_DEFAULT, &v) < 0) throw std::runtime_error("HDF5: read " + name);
        return v;
    }
    std::string read_string(const std::string& name) const
    {
        Dataset d; open(name, d);
        std::string s(Lib::get().H5Tget_size(d.type), '\0');
        if (Lib::get().H5Dread(d.ds, d.type, H5S_ALL, H5S_ALL, H5P_DEFAULT, s.data()) < 0) throw std::runtime_error("HDF5: read " + name);
        while (! s.empty() && s.back() == '\0') s.pop_back();
        return s;
    }
    void read_rational(const std::string& name, int& num, int& den) const
    {
        Dataset d; open(name, d);
        int v[2] = {0, 1};
        if (Lib::get().H5Dread(d.ds, d.type, H5S_ALL, H5S_ALL, H5P_DEFAULT, v) < 0) throw std::runtime_error("HDF5: read " + name);
        num = v[0]; den = v[1];
    }
    std::vector<hsize_t> shape_of(const Dataset& d) const
    {
        auto& L = Lib::get();
        const int rank = L.H5Sget_simple_extent_ndims(d.space);
        std::vector<hsize_t> dims(rank > 0 ? rank : 0);
        if (rank > 0) L.H5Sget_simple_extent_dims(d.space, dims.data(), nullptr);
        return dims;
    }
    std::vector<double> read_vector(const std::string& name) const
    {
        Dataset d; open(name, d);
        const auto dims = shape_of(d);
        hsize_t n = 1;
        for (auto x : dims) n *= x;
        std::vector<double> v(n);
        if (Lib::get().H5Dread(d.ds, Lib::get().native_double, H5S_ALL, H5S_ALL, H5P_DEFAULT, v.data()) < 0) throw std::runtime_error("HDF5: read " + name);
        return v;
    }
    std::vector<double> read_cells(const std::string& name, int nq, std::vector<hsize_t>& shape) const
    {
        Dataset d; open(name, d);
        shape = shape_of(d);
        if (Lib::get().H5Tget_size(d.type) != std::size_t(nq) * sizeof(double)) throw std::runtime_error("HDF5: " + name + " does not hold arrays of " + std::to_string(nq) + " doubles");
        hsize_t n = nq;
        for (auto x : shape) n *= x;
        std::vector<double> v(n);
        if (Lib::get().H5Dread(d.ds, d.type, H5S_ALL, H5S_ALL, H5P_DEFAULT, v.data()) < 0) throw std::runtime_error("HDF5: read " + name);
        return v;
    }
    void read_array(const std::string& name, double* v, int n) const
    {
        Dataset d; open(name, d);
        if (Lib::get().H5Tget_size(d.type) != std::size_t(n) * sizeof(double)) throw std::runtime_error("HDF5: " + name + " is not an array of " + std::to_string(n) + " doubles");
        if (Lib::get().H5Dread(d.ds, d.type, H5S_ALL, H5S_ALL, H5P_DEFAULT, v) < 0) throw std::runtime_error("HDF5: read " + name);
    }
    void read_record(const std::string& name, const Compound& type, void* data) const
    {
        Dataset d; open(name, d);
        if (Lib::get().H5Dread(d.ds, type.id, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0) throw std::runtime_error("HDF5: read " + name);
    }
    hsize_t count_of(const std::string& name) const
    {
        Dataset d; open(name, d);
        const auto n = Lib::get().H5Sget_simple_extent_npoints(d.space);
        return n < 0 ? 0 : hsize_t(n);
    }
    void read_records(const std::string& name, const Compound& type, void* data) const
    {
        Dataset d; open(name, d);
        if (Lib::get().H5Sget_simple_extent_npoints(d.space) <= 0) return;
        if (Lib::get().H5Dread(d.ds, type.id, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0) throw std::runtime_error("HDF5: read " + name);
    }
    H5T_class_t class_of(const std::string& name) const
    {
        Dataset d; open(name, d);
        return Lib::get().H5Tget_class(d.type);
    }
};

#endif // MH_HOST_HAVE_HDF5

// ---- mara::schedule_t (src/app_schedule.hpp:55-160) -----------------------------------------------------------------------
struct schedule_t
{
    struct task_t { std::string name; int num_times_performed = 0; double last_performed = 0.0; bool is_due = false; };
    std::map<std::string, task_t> tasks;

    task_t& at(const std::string& n) { auto it = tasks.find(n); if (it == tasks.end()) throw std::out_of_range("no task scheduled with the name " + n); return it->second; }
    const task_t& at(const std::string& n) const { auto it = tasks.find(n); if (it == tasks.end()) throw std::out_of_range("no task scheduled with the name " + n); return it->second; }
    void create_and_mark_as_due(const std::string& n) { tasks[n] = {n, 0, 0.0, false}; mark_as_due(n); }
    void mark_as_due(const std::string& n, double increase_last_performed_by = 0.0) { at(n).is_due = true; at(n).last_performed += increase_last_performed_by; }
    void mark_as_completed(const std::string& n) { at(n).is_due = false; at(n).num_times_performed += 1; }
    bool is_due(const std::string& n) const { return at(n).is_due; }
    // next_schedule of the drivers (subprog_cloud.cpp:720-733): a task is due again once `interval` has passed since it was last due
    void advance(const std::string& n, double time, double interval) { if (time - at(n).last_performed >= interval) mark_as_due(n, interval); }
};

inline void write_schedule(const Node& group, const schedule_t& s)
{
    for (const auto& kv : s.tasks)
    {
        Node t = group.require_group(kv.first);
        t.write("name", kv.second.name);
        t.write("num_times_performed", kv.second.num_times_performed);
        t.write("last_performed", kv.second.last_performed);
    }
}

inline schedule_t read_schedule(const Node& group)
{
    schedule_t s;
    for (const auto& name : group.names())
    {
        Node t = group.open_group(name);
        schedule_t::task_t task;
        task.name = name;
        task.num_times_performed = t.read_int("num_times_performed");
        task.last_performed = t.read_double("last_performed");
        s.tasks[name] = task;                                   // is_due is not stored: false after a restart, as upstream
    }
    return s;
}

inline void write_config(const Node& group, const mara::config_t& cfg)
{
    for (const auto& kv : cfg.items())
    {
        switch (kv.second.index())
        {
            case 0: group.write(kv.first, std::get<int>(kv.second)); break;
            case 1: group.write(kv.first, std::get<double>(kv.second)); break;
            case 2: group.write(kv.first, std::get<std::string>(kv.second)); break;
        }
    }
}

// items of a stored run configuration that the template knows, typed by the template (mara::read_config + config_t::update)
inline void read_config_into(const Node& group, mara::config_t& cfg)
{
    for (const auto& name : group.names())
    {
        if (! cfg.has(name)) continue;
        switch (cfg.type_of(name))
        {
            case 0: cfg.item(name, group.read_int(name)); break;
            case 1: cfg.item(name, group.read_double(name)); break;
            case 2: cfg.item(name, group.read_string(name)); break;
        }
    }
}

// mara::format_tree_index (app_serialize_tree.hpp:74-90): "level:ii-jj", coordinates zero-padded to 1 + log10(2^level) digits
// (known answers of app_test.cpp:377-384: level 3 -> "3:5-6", level 5 -> "5:01-16", level 8 -> "8:001-002")
inline std::string format_tree_index(int level, int i, int j)
{
    const int width = int(1 + std::log10(double(1 << level)));
    char buf[64];
    std::snprintf(buf, sizeof buf, "%d:%0*d-%0*d", level, width, i, width, j);
    return buf;
}

inline std::string numbered_filename(const std::string& prefix, int count, const std::string& ext)
{
    char buf[1024];
    std::snprintf(buf, sizeof buf, "%s.%04d.%s", prefix.c_str(), count, ext.c_str());     // mara::create_numbered_filename app_serialize.hpp:170-175
    return buf;
}

} // namespace h5io
