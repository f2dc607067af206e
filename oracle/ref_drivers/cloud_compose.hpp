/**
 * TEST INFRASTRUCTURE — `mara cloud` composed from the reference's own headers (geometry, boundary conditions, advance, the problem
 * set-up of CloudProblem::new_solution), shared by the reference-side drivers cloud_ref.cpp (golden vectors, long runs) and
 * integration_ref.cpp (the boundary compiled against the reference's types). See cloud_ref.cpp for the citations.
 */
#pragma once
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
#include "app_parallel.hpp"

namespace cloud_compose {

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

// `evaluate`: nd::to_shared(), or the reference's threaded twin mara::evaluate_on<N>() (app_parallel.hpp:72-103) - upstream pipes
// `evaluate_on<MARA_PREFERRED_THREAD_COUNT>()` at exactly these six places (:525-533, :582) and keeps nd::to_shared() for the nozzle row and
// the PLM gradients (:566)
template<typename U, typename Evaluator>
static auto advance(const setup_t& S, U u0, double time, mara::unit_time<double> dt, std::vector<double>* inflow_dump, Evaluator evaluate)
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

    auto rc  = cell_centroids(S.rv, S.qv) | evaluate;
    auto dv  = cell_volumes(S.rv, S.qv) | evaluate;
    auto dAr = radial_face_areas(S.rv, S.qv) | evaluate;
    auto dAq = polar_face_areas(S.rv, S.qv) | evaluate;

    auto p0 = u0 / dv | nd::map(c2p) | evaluate;
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
        return (u0 + (lr + lq + s0) * dt) | evaluate;
    }
    auto lr = p0 | extend_bc | extrapolate_plm(0) | flux_on(0)                       | nd::multiply(-dAr) | nd::difference_on_axis(0);
    auto lq = p0 |             extrapolate_plm(1) | flux_on(1) | nd::extend_zeros(1) | nd::multiply(-dAq) | nd::difference_on_axis(1);
    return (u0 + (lr + lq + s0) * dt) | evaluate;
}

template<typename U>
static auto advance(const setup_t& S, U u0, double time, mara::unit_time<double> dt, std::vector<double>* inflow_dump)
{
    return advance(S, u0, time, dt, inflow_dump, nd::to_shared());
}

// f(evaluator) with the reference's threaded evaluator on `threads` threads (a template parameter upstream: MARA_PREFERRED_THREAD_COUNT = 12)
template<class F>
static auto with_upstream_evaluator(int threads, F f)
{
    switch (threads)
    {
        case 1:  return f(mara::evaluate_on<1>());
        case 2:  return f(mara::evaluate_on<2>());
        case 4:  return f(mara::evaluate_on<4>());
        case 8:  return f(mara::evaluate_on<8>());
        case 12: return f(mara::evaluate_on<12>());
        case 16: return f(mara::evaluate_on<16>());
        case 32: return f(mara::evaluate_on<32>());
        default: return f(nd::to_shared());
    }
}



// The sub-program's models, units, vertices and initial state with its option defaults: fills S, returns the cell-integrated conserved
// array and (by reference) the reference length and mass the diagnostics need.
static auto make_cloud_problem(setup_t& S, int nr, double num_decades, double& ref_length_out, double& ref_mass_out)
{
    // option defaults, subprog_cloud.cpp:60-87
    const double inner_radius = 3e08, cloud_cutoff = 3e10, cloud_mass = 2e-2, density_index = 2.0, density_index2 = 6.0;
    const double jet_delay_time = 1.0, jet_total_energy = 1e50, jet_duration = 1.0, jet_gamma_beta = 10.0;
    const double jet_opening_angle = 0.1, jet_structure_exp = 2.0;

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

    ref_length_out = ref_length;
    ref_mass_out = ref_mass;
    return u;
}

} // namespace cloud_compose
