/**
 * TEST INFRASTRUCTURE — reference-side oracle driver for BASELINE config 4
 * (`mara cloud`): 2-D axisymmetric spherical-polar SRHD jet-cloud.
 *
 * The sub-program itself needs HDF5 (subprog_cloud.cpp includes core_hdf5.hpp),
 * so this driver calls the same reference header functions, in the same order,
 * as CloudProblem::new_solution (subprog_cloud.cpp:610-662), the geometry
 * helpers (:260-290), the boundary conditions (:466-509), advance (:511-584)
 * and next_solution (:676-697), with gamma = 4/3 (:52) and the option defaults
 * of :60-87.
 *
 * usage: cloud_ref <nr> <num_decades> <rk_order> <reconstruct_method> <plm_theta> <nsteps> <outprefix>
 * writes <outprefix>.{rv,qv,u0,un,inflow,meta,diag_fields,diag_columns,diag_meta}.f64
 *   inflow: [nsteps][nq][5] primitives of the inner radial ghost cells at each step's start time
 *   meta  : dt, temperature_floor
 *   diag_*: CloudProblem::make_diagnostic_fields (subprog_cloud.cpp:334-433) of the final state, composed from the same header
 *           functions (post_shock_locator.hpp, nd::freeze_axis, ...): diag_fields [5][nr][nq] = mass_density, gas_pressure,
 *           specific_entropy, radial_gamma_beta, radial_energy_flow; diag_columns [15][nq] in the order of diagnostic_fields_t
 *           (:147-161: total_energy, solid_angle, the four shock radii, postshock gamma, power, power02..64, power_max);
 *           diag_meta = time (s), reference length, mass, time
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <numeric>
#include <vector>
#include <string>
#include "core_ndarray.hpp"
#include "core_ndarray_ops.hpp"
#include "core_dimensional.hpp"
#include "core_sequence.hpp"
#include "core_geometric.hpp"
#include "core_rational.hpp"   // physics_srhd.hpp:443 formats a double through mara::to_string(rational_number_t)
#include "physics_srhd.hpp"
#include "model_atmosphere.hpp"
#include "model_jet_nozzle.hpp"
#include "post_shock_locator.hpp"

static const double gamma_law = 4. / 3;
static const double light_speed_cgs = 2.998e10;
static const double solar_mass_cgs = 1.989e33;

using prim_t = mara::srhd::primitive_t;
using rv_t = nd::shared_array<mara::unit_length<double>, 1>;
using qv_t = nd::shared_array<double, 1>;

template<typename A> static auto vsin(A a) { return a | nd::map([] (auto x) { return std::sin(x); }); }
template<typename A> static auto vcos(A a) { return a | nd::map([] (auto x) { return std::cos(x); }); }

static auto radial_face_areas(rv_t rv, qv_t qv)
{
    auto [r, q] = nd::meshgrid(rv, qv);
    auto rc = r | nd::midpoint_on_axis(1);
    auto dm = -vcos(q) | nd::difference_on_axis(1);
    return rc * rc * dm * 2 * M_PI;
}
static auto polar_face_areas(rv_t rv, qv_t qv)
{
    auto [r, q] = nd::meshgrid(rv, qv);
    auto dr = r | nd::difference_on_axis(0);
    auto rc = r | nd::midpoint_on_axis(0);
    auto qc = q | nd::midpoint_on_axis(0);
    return rc * dr * vsin(qc) * 2 * M_PI;
}
static auto cell_volumes(rv_t rv, qv_t qv)
{
    auto [r, q] = nd::meshgrid(rv, qv);
    auto dv = r | nd::map([] (auto r) { return r * r * r; }) | nd::difference_on_axis(0) | nd::midpoint_on_axis(1);
    auto dm = -vcos(q) | nd::difference_on_axis(1) | nd::midpoint_on_axis(0);
    return dv * dm * 2 * M_PI / 3.0;
}
static auto cell_centroids(rv_t rv, qv_t qv)
{
    return nd::cartesian_product(rv | nd::midpoint_on_axis(0), qv | nd::midpoint_on_axis(0));
}

static double plm1(double ul, double u0, double ur, double theta)   // subprog_cloud.cpp:450-464
{
    using std::min; using std::fabs;
    auto sgn = [] (double x) { return std::copysign(1, x); };
    auto a = theta * (u0 - ul);
    auto b = 0.5 * (ur - ul);
    auto c = theta * (ur - u0);
    return 0.25 * fabs(sgn(a) + sgn(b)) * (sgn(a) + sgn(c)) * min(min(fabs(a), fabs(b)), fabs(c));
}

struct setup_t
{
    rv_t rv; qv_t qv;
    mara::jet_nozzle_model jet;
    double ref_time, ref_density;
    double temperature_floor = 1e-8, theta = 1.2;
    int method = 2;
};

template<typename U>
static auto advance(const setup_t& S, U u0, double time, mara::unit_time<double> dt, std::vector<double>* inflow_dump)
{
    auto source_terms = [] (auto primitive, auto position)
    {
        return primitive.spherical_geometry_source_terms(std::get<0>(position).value, std::get<1>(position), gamma_law);
    };
    auto c2p = [tf=S.temperature_floor] (auto U_) { return mara::srhd::recover_primitive(U_, gamma_law, tf); };

    auto polar_cells = S.qv | nd::midpoint_on_axis(0);
    auto t_seconds = time * S.ref_time;
    auto inflow_function = [jet=S.jet, t=t_seconds, rd=S.ref_density] (double q)
    {
        auto u = jet.gamma_beta(q, t) + jet.gamma_beta(M_PI - q, t);
        auto d = jet.density_at_base() / rd;
        return prim_t().with_mass_density(d).with_gamma_beta_1(u);
    };
    auto inflow_row = polar_cells | nd::map(inflow_function) | nd::to_shared();
    if (inflow_dump)
        for (auto p : inflow_row) for (int q = 0; q < 5; ++q) inflow_dump->push_back(p[q]);

    auto extend_bc = [inflow_row, polar_cells] (auto array)
    {
        auto inner = inflow_row | nd::reshape(1, polar_cells.size()) | nd::concat(array);
        return inner | nd::concat(array | nd::select_final(1, 0));
    };

    auto rc  = cell_centroids(S.rv, S.qv) | nd::to_shared();
    auto dv  = cell_volumes(S.rv, S.qv) | nd::to_shared();
    auto dAr = radial_face_areas(S.rv, S.qv) | nd::to_shared();
    auto dAq = polar_face_areas(S.rv, S.qv) | nd::to_shared();

    auto p0 = u0 / dv | nd::map(c2p) | nd::to_shared();
    auto s0 = nd::zip(p0, rc) | nd::apply(source_terms) | nd::multiply(dv);

    auto flux_on = [] (std::size_t axis)
    {
        return [axis] (auto lr)
        {
            auto nh = mara::unit_vector_t::on_axis(axis);
            return lr | nd::apply([nh] (prim_t l, prim_t r) { return mara::srhd::riemann_hlle(l, r, nh, gamma_law); });
        };
    };
    auto extrapolate_pcm = [] (std::size_t axis) { return nd::zip_adjacent2_on_axis(axis); };
    auto extrapolate_plm = [theta=S.theta] (std::size_t axis)
    {
        return [axis, theta] (auto P)
        {
            auto L = nd::select_axis(axis).from(0).to(1).from_the_end();
            auto R = nd::select_axis(axis).from(1).to(0).from_the_end();
            auto G = P
            | nd::zip_adjacent3_on_axis(axis)
            | nd::apply(mara::lift([theta] (double a, double b, double c) { return plm1(a, b, c, theta); }))
            | nd::extend_zeros(axis)
            | nd::to_shared();
            return nd::zip((P | L) + (G | L) * 0.5, (P | R) - (G | R) * 0.5);
        };
    };

    if (S.method == 1)
    {
        auto lr = p0 | extend_bc | extrapolate_pcm(0) | flux_on(0)                       | nd::multiply(-dAr) | nd::difference_on_axis(0);
        auto lq = p0 |             extrapolate_pcm(1) | flux_on(1) | nd::extend_zeros(1) | nd::multiply(-dAq) | nd::difference_on_axis(1);
        return (u0 + (lr + lq + s0) * dt) | nd::to_shared();
    }
    auto lr = p0 | extend_bc | extrapolate_plm(0) | flux_on(0)                       | nd::multiply(-dAr) | nd::difference_on_axis(0);
    auto lq = p0 |             extrapolate_plm(1) | flux_on(1) | nd::extend_zeros(1) | nd::multiply(-dAq) | nd::difference_on_axis(1);
    return (u0 + (lr + lq + s0) * dt) | nd::to_shared();
}

int main(int argc, char** argv)
{
    if (argc != 8) return 1;
    int nr = std::atoi(argv[1]);
    double num_decades = std::atof(argv[2]);
    int rk = std::atoi(argv[3]);
    setup_t S;
    S.method = std::atoi(argv[4]);
    S.theta = std::atof(argv[5]);
    int nsteps = std::atoi(argv[6]);
    std::string prefix = argv[7];

    // option defaults, subprog_cloud.cpp:60-87
    const double inner_radius = 3e08, cloud_cutoff = 3e10, cloud_mass = 2e-2, density_index = 2.0, density_index2 = 6.0;
    const double jet_delay_time = 1.0, jet_total_energy = 1e50, jet_duration = 1.0, jet_gamma_beta = 10.0;
    const double jet_opening_angle = 0.1, jet_structure_exp = 2.0, cfl_number = 0.4;

    auto envelop = mara::cloud_and_envelop_model().with_inner_radius(inner_radius).with_cloud_index(density_index);
    auto atmosphere = mara::power_law_atmosphere_model()
    .with_inner_radius(inner_radius).with_cutoff_radius(cloud_cutoff).with_inner_index(density_index)
    .with_outer_index(density_index2).with_total_mass(cloud_mass * solar_mass_cgs);
    S.jet = mara::jet_nozzle_model()
    .with_inner_radius(inner_radius).with_total_energy(jet_total_energy).with_jet_duration(jet_duration)
    .with_structure_exponent(jet_structure_exp).with_opening_angle(jet_opening_angle).with_lorentz_factor(jet_gamma_beta);

    const double ref_length = atmosphere.r0;
    const double ref_mass = atmosphere.total_mass();
    S.ref_time = atmosphere.r0 / light_speed_cgs;
    S.ref_density = ref_mass / std::pow(ref_length, 3);

    S.rv = nd::linspace(0.0, num_decades, int(num_decades * nr) + 1)
    | nd::map([] (auto y) { return mara::make_length(std::pow(10.0, y)); }) | nd::to_shared();
    S.qv = nd::linspace(0.0, M_PI, nr + 1) | nd::to_shared();

    auto initial_p = [envelop, ref_length, rd=S.ref_density, jet_delay_time] (auto r, auto q)
    {
        auto r_cm = r.value * ref_length;
        auto temperature = 1e-6;
        auto density = envelop.density_at(r_cm, jet_delay_time) / rd;
        auto gamma_beta = envelop.gamma_beta_at(r_cm, jet_delay_time);
        return prim_t().with_mass_density(density).with_gas_pressure(density * temperature).with_gamma_beta_1(gamma_beta);
    };
    auto dv = cell_volumes(S.rv, S.qv);
    auto u = cell_centroids(S.rv, S.qv)
    | nd::apply(initial_p)
    | nd::map([] (prim_t p) { return p.to_conserved_density(gamma_law); })
    | nd::multiply(dv)
    | nd::to_shared();
    auto u_init = u;

    auto dr_min = S.rv | nd::difference_on_axis(0) | nd::read_index(0);
    auto dt = dr_min / mara::make_velocity(1.0) * cfl_number;
    double time = 0.0;
    std::vector<double> inflow;

    for (int n = 0; n < nsteps; ++n)
    {
        if (rk == 1)
        {
            u = advance(S, u, time, dt, &inflow);
        }
        else
        {
            auto s1 = advance(S, u, time, dt, &inflow);
            auto s2 = advance(S, s1, time, dt, nullptr);         // both stages see the step-start time (:468-473, :524)
            u = (u * 0.5 + s2 * (1 - 0.5)) | nd::to_shared();
        }
        time += dt.value;
    }

    auto dump = [&] (const char* ext, const void* data, std::size_t bytes)
    {
        FILE* f = std::fopen((prefix + "." + ext + ".f64").c_str(), "wb");
        std::fwrite(data, 1, bytes, f);
        std::fclose(f);
    };
    static_assert(sizeof(mara::srhd::conserved_t) == 40, "layout");
    dump("rv", S.rv.data(), S.rv.size() * sizeof(double));
    dump("qv", S.qv.data(), S.qv.size() * sizeof(double));
    dump("u0", u_init.data(), u_init.size() * 40);
    dump("un", u.data(), u.size() * 40);
    dump("inflow", inflow.data(), inflow.size() * sizeof(double));
    double meta[2] = {dt.value, S.temperature_floor};
    dump("meta", meta, sizeof meta);

    // ---- make_diagnostic_fields, subprog_cloud.cpp:334-433 (unit_system_t :177-195)
    {
        using namespace std::placeholders;
        const double u_length = ref_length, u_mass = ref_mass, u_time = S.ref_time;
        const double u_energy = u_mass * std::pow(light_speed_cgs, 2);
        const double u_mass_density = u_mass / std::pow(u_length, 3);
        const double u_energy_density = u_energy / std::pow(u_length, 3);
        const double u_power = u_energy / u_time;
        auto dv           = cell_volumes(S.rv, S.qv);
        auto dAr          = radial_face_areas(S.rv, S.qv);
        auto rhat         = mara::unit_vector_t::on_axis_1();
        auto cons_to_prim = std::bind(mara::srhd::recover_primitive, _1, gamma_law, S.temperature_floor);
        auto radial_cells = S.rv | nd::midpoint_on_axis(0);
        auto primitive    = u | nd::divide(dv) | nd::map(cons_to_prim);
        const std::size_t nq = S.qv.size() - 1, nrr = S.rv.size() - 1;
        std::vector<std::vector<double>> col(15, std::vector<double>(nq));
        for (std::size_t j = 0; j < nq; ++j)
        {
            auto pj = primitive | nd::freeze_axis(1).at_index(j) | nd::to_shared();
            auto uj = u         | nd::freeze_axis(1).at_index(j);
            auto Aj = dAr | nd::freeze_axis(1).at_index(j) | nd::midpoint_on_axis(0);
            auto Lj = pj | nd::map([rhat] (auto p) { return p.flux(rhat, gamma_law)[4]; })
            | nd::multiply(Aj)
            | nd::multiply(u_power)
            | nd::map([] (auto L) { return L.value; });

            auto midpoint_index   = mara::find_shock_index(pj, gamma_law)[0];
            auto upstream_index   = mara::find_index_of_pressure_plateau_ahead(pj, midpoint_index);
            auto pressure_index   = mara::find_index_of_maximum_pressure_behind(pj, midpoint_index);
            auto luminosity_index = mara::find_index_of_maximum_behind(Lj, midpoint_index);
            auto i02 = midpoint_index >  2 ? midpoint_index -  2 : 0;
            auto i04 = midpoint_index >  4 ? midpoint_index -  4 : 0;
            auto i08 = midpoint_index >  8 ? midpoint_index -  8 : 0;
            auto i16 = midpoint_index > 16 ? midpoint_index - 16 : 0;
            auto i32 = midpoint_index > 32 ? midpoint_index - 32 : 0;
            auto i64 = midpoint_index > 64 ? midpoint_index - 64 : 0;

            col[0][j]  = uj | nd::map([] (auto u_) { return u_[4].value; }) | nd::multiply(u_energy) | nd::sum();
            col[1][j]  = dAr(0, j) / S.rv(0) / S.rv(0);
            col[2][j]  = radial_cells(midpoint_index).value * u_length;
            col[3][j]  = radial_cells(upstream_index).value * u_length;
            col[4][j]  = radial_cells(pressure_index).value * u_length;
            col[5][j]  = radial_cells(luminosity_index).value * u_length;
            col[6][j]  = primitive(pressure_index, j).lorentz_factor();
            col[7][j]  = Lj(pressure_index);
            col[8][j]  = Lj(i02);
            col[9][j]  = Lj(i04);
            col[10][j] = Lj(i08);
            col[11][j] = Lj(i16);
            col[12][j] = Lj(i32);
            col[13][j] = Lj(i64);
            col[14][j] = Lj(luminosity_index);
        }
        auto specific_entropy   = primitive | nd::map(std::bind(&prim_t::specific_entropy, _1, gamma_law)) | nd::to_shared();
        auto gas_pressure       = primitive | nd::map(std::mem_fn(&prim_t::gas_pressure)) | nd::multiply(u_energy_density) | nd::to_shared();
        auto mass_density       = primitive | nd::map(std::mem_fn(&prim_t::mass_density)) | nd::multiply(u_mass_density) | nd::to_shared();
        auto radial_gamma_beta  = primitive | nd::map(std::mem_fn(&prim_t::gamma_beta_1)) | nd::to_shared();
        auto radial_energy_flow = primitive
        | nd::map([rhat] (auto p) { return p.flux(rhat, gamma_law); })
        | nd::multiply(dAr | nd::select_axis(0).from(0).to(1).from_the_end())
        | nd::map([] (auto L) { return L[4].value; })
        | nd::multiply(u_power)
        | nd::to_shared();
        std::vector<double> fields;
        for (auto v : mass_density) fields.push_back(v);
        for (auto v : gas_pressure) fields.push_back(v);
        for (auto v : specific_entropy) fields.push_back(v);
        for (auto v : radial_gamma_beta) fields.push_back(v);
        for (auto v : radial_energy_flow) fields.push_back(v);
        dump("diag_fields", fields.data(), fields.size() * sizeof(double));
        std::vector<double> columns;
        for (auto& c : col) for (double v : c) columns.push_back(v);
        dump("diag_columns", columns.data(), columns.size() * sizeof(double));
        double dmeta[4] = {time * u_time, u_length, u_mass, u_time};
        dump("diag_meta", dmeta, sizeof dmeta);
        (void) nrr;
    }
    return 0;
}
