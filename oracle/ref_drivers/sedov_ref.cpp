/**
 * TEST INFRASTRUCTURE — reference-side oracle driver for BASELINE config 1
 * (`mara sedov newtonian=1`). The sub-program itself cannot be built here
 * without HDF5 (subprog_sedov.cpp includes core_hdf5.hpp), so this driver
 * calls the same reference header functions in the same order as
 * SedovProblem<mara::euler>::new_solution / next_solution
 * (subprog_sedov.cpp:353-421, BCs :217-250, geometry :166-181), with
 * gamma = 4/3 and CFL = 0.4 as #defined at :48-49.
 *
 * usage: sedov_ref <nr> <outer_radius> <nsteps> <out_vertices.f64> <out_u0.f64> <out_uN.f64> [srhd]
 * (the optional last argument selects mara::srhd, the sub-program's default system, instead of mara::euler)
 * Also writes <out_uN.f64>.diag: make_diagnostic_fields / compute_time_series_data of the final state (subprog_sedov.cpp:252-308,
 * composed from the same header functions): [4][nz] specific_entropy, gas_pressure, mass_density, radial velocity or gamma-beta;
 * then shock, downstream, upstream index (as doubles); then time, shock_radius, shock_radius_upstream, shock_radius_downstream,
 * shock_radius_interpolated, shock_velocity.
 */
#include <cstdio>
#include <string>
#include <vector>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <numeric>
#include <functional>
#include "core_ndarray.hpp"
#include "core_ndarray_ops.hpp"
#include "core_dimensional.hpp"
#include "core_sequence.hpp"
#include "core_geometric.hpp"
#include "core_rational.hpp"
#include "physics_euler.hpp"
#include "physics_srhd.hpp"
#include "post_shock_locator.hpp"
#include "math_polynomial.hpp"

static const double gamma_law = 4. / 3;
static const double cfl = 0.4;

static double radial_u(const mara::srhd::primitive_t& p) { return p.gamma_beta_1(); }          // subprog_sedov.cpp:76-84
static double radial_u(const mara::euler::primitive_t& p) { return p.velocity_1(); }
static double shock_velocity(const mara::srhd::primitive_t& p1, const mara::srhd::primitive_t& p2)   // :96-105
{
    auto d1 = p1.mass_density(), d2 = p2.mass_density(), u1 = p1.gamma_beta_1(), u2 = p2.gamma_beta_1();
    auto g1 = p1.lorentz_factor(), g2 = p2.lorentz_factor();
    return (d2 * u2 - d1 * u1) / (d2 * g2 - d1 * g1);
}
static double shock_velocity(const mara::euler::primitive_t& p1, const mara::euler::primitive_t& p2)   // :107-114
{
    auto d1 = p1.mass_density(), d2 = p2.mass_density(), v1 = p1.velocity_1(), v2 = p2.velocity_1();
    return (d2 * v2 - d1 * v1) / (d2 - d1);
}
static auto negate_radial(const mara::euler::primitive_t& p) { return p.with_velocity_1(-p.velocity_1()); }
static auto negate_radial(const mara::srhd::primitive_t& p) { return p.with_gamma_beta_1(-p.gamma_beta_1()); }

template<typename V> static auto face_areas(V vertices)
{
    return vertices | nd::map([] (auto r) { return mara::make_area(r * r); });
}
template<typename V> static auto cell_volumes(V vertices)
{
    auto shell = [] (double r0, double r1) { return mara::make_volume((std::pow(r1, 3) - std::pow(r0, 3)) / 3); };
    return vertices | nd::zip_adjacent2_on_axis(0) | nd::apply(shell);
}

template<typename HydroSystem>
static int run(int argc, char** argv)
{
    using prim_t = typename HydroSystem::primitive_t;
    using cons_t = typename HydroSystem::conserved_t; // volume-integrated
    int nr = std::atoi(argv[1]);
    double outer_radius = std::atof(argv[2]);
    int nsteps = std::atoi(argv[3]);

    auto decades = std::log10(outer_radius);
    auto vertices = nd::linspace(-0.5, decades, int(decades * nr) + 1)
    | nd::map([] (auto y) { return std::pow(10.0, y); })
    | nd::to_shared();

    auto initial_p = [] (double r)
    {
        return prim_t()
        .with_mass_density(r < 1.0 ? 1.0 : std::pow(r, -0.0))
        .with_gas_pressure(r < 1.0 ? 1.0 : std::pow(r, -0.0) * 1e-6);
    };
    auto to_cons = [] (prim_t p) { return p.to_conserved_density(gamma_law); };
    auto c2p = [] (auto U) { return HydroSystem::recover_primitive(U, gamma_law, 0.0); };
    auto nh = mara::unit_vector_t::on_axis_1();
    auto riemann = [nh] (prim_t l, prim_t r) { return HydroSystem::riemann_hlle(l, r, nh, gamma_law); };
    auto source = [] (prim_t p, double r) { return p.spherical_geometry_source_terms_radial(r, gamma_law); };
    auto reflect = [] (prim_t p) { return negate_radial(p); };

    auto xc = vertices | nd::midpoint_on_axis(0);
    auto u = xc | nd::map(initial_p) | nd::map(to_cons) | nd::multiply(cell_volumes(vertices)) | nd::to_shared();
    auto u_init = u;

    double time = 0.0;
    for (int n = 0; n < nsteps; ++n)
    {
        time += cfl * (vertices(1) - vertices(0));
        auto dr_min = vertices | nd::difference_on_axis(0) | nd::read_index(0);
        auto dt = mara::make_time(cfl * dr_min);
        auto dv = cell_volumes(vertices) | nd::to_shared();
        auto da = face_areas(vertices);
        auto rc = vertices | nd::midpoint_on_axis(0);
        auto p0 = u / dv | nd::map(c2p) | nd::to_shared();
        auto s0 = nd::zip(p0, rc) | nd::apply(source) | nd::multiply(dv);
        auto pe = (p0 | nd::select_first(1, 0) | nd::map(reflect)) | nd::concat(p0) | nd::concat(p0 | nd::select_final(1, 0));
        auto fl = nd::zip(pe | nd::select_axis(0).from(0).to(1).from_the_end(),
                          pe | nd::select_axis(0).from(1).to(0).from_the_end()) | nd::apply(riemann);
        auto l0 = fl | nd::multiply(-da) | nd::difference_on_axis(0);
        u = (u + (l0 + s0) * dt) | nd::to_shared();
    }

    auto dump = [] (const char* name, const void* data, std::size_t bytes)
    {
        FILE* f = std::fopen(name, "wb");
        std::fwrite(data, 1, bytes, f);
        std::fclose(f);
    };
    static_assert(sizeof(cons_t) == 40, "layout");
    dump(argv[4], vertices.data(), vertices.size() * sizeof(double));
    dump(argv[5], u_init.data(), u_init.size() * sizeof(cons_t));
    dump(argv[6], u.data(), u.size() * sizeof(cons_t));
    {
        using namespace std::placeholders;
        auto primitive = u | nd::divide(cell_volumes(vertices)) | nd::map(c2p) | nd::to_shared();
        std::vector<double> out;
        for (auto p : primitive) out.push_back(p.specific_entropy(gamma_law));
        for (auto p : primitive) out.push_back(p.gas_pressure());
        for (auto p : primitive) out.push_back(p.mass_density());
        for (auto p : primitive) out.push_back(radial_u(p));
        auto shock_index      = mara::find_shock_index(primitive, gamma_law)[0];
        auto downstream_index = mara::find_index_of_maximum_pressure_behind(primitive, shock_index);
        auto upstream_index   = mara::find_index_of_pressure_plateau_ahead(primitive, shock_index);
        auto rc = vertices | nd::midpoint_on_axis(0);
        auto vc = primitive | nd::map([] (auto p) { return radial_u(p); });
        auto find_vertex = [rc, vc] (auto i) { return mara::parabola_vertex(rc(i - 1), rc(i), rc(i + 1), vc(i - 1), vc(i), vc(i + 1)).first; };
        out.push_back(double(shock_index)); out.push_back(double(downstream_index)); out.push_back(double(upstream_index));
        out.push_back(time);
        out.push_back(vertices(shock_index));
        out.push_back(rc(upstream_index));
        out.push_back(rc(downstream_index));
        out.push_back(downstream_index >= 1 && downstream_index + 1 < rc.size() ? find_vertex(downstream_index) : std::nan(""));   // upstream reads out of range there
        out.push_back(shock_velocity(primitive(upstream_index), primitive(downstream_index)));
        dump((std::string(argv[6]) + ".diag").c_str(), out.data(), out.size() * sizeof(double));
    }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc != 7 && argc != 8) return 1;
    return argc == 8 ? run<mara::srhd>(argc, argv) : run<mara::euler>(argc, argv);
}
