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
 */
#include <cstdio>
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

static const double gamma_law = 4. / 3;
static const double cfl = 0.4;

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

    for (int n = 0; n < nsteps; ++n)
    {
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
    return 0;
}

int main(int argc, char** argv)
{
    if (argc != 7 && argc != 8) return 1;
    return argc == 8 ? run<mara::srhd>(argc, argv) : run<mara::euler>(argc, argv);
}
