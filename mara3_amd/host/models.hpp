// Host-side scalar models used by the `cloud` sub-program's initial condition, unit system and nozzle
// boundary condition. They are evaluated once per cell at start-up (IC) or once per polar cell per step
// (nozzle), never on the per-zone hot path, and they are libm-heavy (pow/exp/log10), so - like the
// reference - they stay on the host (model_jet_nozzle / model_atmosphere "keep on CPU").
// Formulas and evaluation order follow
//   mara::jet_nozzle_model            src/model_jet_nozzle.hpp:41-100
//   mara::power_law_atmosphere_model  src/model_atmosphere.hpp:97-152
//   mara::cloud_and_envelop_model     src/model_atmosphere.hpp:155-281 (secant solver :56-78)
// so that the initial state and the inflow row are bit-identical to the reference's (checked against the
// reference-generated fixtures in tests/test_gpu_host_subprograms.py).
#pragma once
#include <cmath>
#include <stdexcept>

namespace model {

constexpr double light_speed_jet = 3e10;        // jet_nozzle_model::light_speed_cgs (:43)
constexpr double solar_mass = 1.989e33;         // cloud_and_envelop_model::solar_mass (:272)
constexpr double light_speed = 2.998e10;        // cloud_and_envelop_model::light_speed (:273)

struct jet_nozzle
{
    double Ej = 1.0, G0 = 2.0, tj = 1.0, qj = 0.1, as = 2.0, r0 = 1.0;

    double density_at_base() const
    {
        return Ej / (2 * M_PI * std::pow(G0 * r0 * qj, 2) * tj * std::pow(light_speed_jet, 3));
    }
    double gamma_beta(double q, double t) const
    {
        return G0 * std::exp(-0.5 * std::pow(q / qj, as)) * std::exp(-0.5 * t / tj);
    }
};

struct power_law_atmosphere
{
    double f0 = 1.0, r0 = 1.0, rc = 1e2, n1 = 2.0, n2 = 6.0;

    double density_at(double r) const
    {
        return r <= rc ? f0 * std::pow(r / r0, -n1) : density_at(rc) * std::pow(r / rc, -n2);
    }
    double mass_within_cutoff() const
    {
        return n1 == 3.0
        ? 4 * M_PI * (density_at(rc) * std::pow(rc, 3) * std::log(rc / r0))
        : 4 * M_PI * (density_at(rc) * std::pow(rc, 3) - density_at(r0) * std::pow(r0, 3)) / (3 - n1);
    }
    double mass_beyond_cutoff() const
    {
        if (n2 <= 3.0) throw std::invalid_argument("power_law_atmosphere: outer index (n2) must be greater than 3");
        return 4 * M_PI * density_at(rc) * std::pow(rc, 3) / (n2 - 3);
    }
    double total_mass() const { return mass_within_cutoff() + mass_beyond_cutoff(); }
    power_law_atmosphere with_total_mass(double new_total_mass) const
    {
        auto result = *this;
        result.f0 = new_total_mass / total_mass();
        return result;
    }
};

template<typename Function>
double solve_secant(Function f, double x1, double x2, double tolerance)
{
    double y1 = f(x1);
    double y2 = f(x2);
    while (std::abs(y2) > tolerance)
    {
        const double x_next = x2 - y2 * (x2 - x1) / (y2 - y1);
        const double y_next = f(x_next);
        x1 = x2;
        y1 = y2;
        x2 = x_next;
        y2 = y_next;
    }
    return x2;
}

struct cloud_and_envelop
{
    double inner_radius = 3e8;
    double envelop_mass = 0.005 * solar_mass;
    double u1 = 4.0, m1 = 1e26, psi = 0.25, cloud_index = 2.0;

    double gamma_beta(double m) const { return u1 * std::pow(m / m1, -psi); }
    double velocity(double m) const
    {
        const double u = gamma_beta(m);
        return u / std::sqrt(1.0 + u * u) * light_speed;
    }
    double dudm(double m) const { return -psi / m * gamma_beta(m); }
    double radius(double m, double t) const { return velocity(m) * t; }
    double density(double m, double t) const
    {
        const double gamma_squared = 1.0 + std::pow(gamma_beta(m), 2);
        const double beta = velocity(m) / light_speed;
        return gamma_squared * beta / (4 * M_PI * std::pow(radius(m, t), 3)) / std::abs(dudm(m));
    }
    double cloud_gamma_beta() const
    {
        const double beta = velocity(envelop_mass) / light_speed;
        return beta / std::sqrt(1.0 - beta * beta);
    }
    double cloud_outer_boundary(double t) const { return velocity(envelop_mass) * t; }
    double envelop_outer_boundary(double t) const { return radius(m1, t); }
    double mass_coordinate(double r, double t) const
    {
        auto f = [this, r, t] (double m) { return std::log10(r) - std::log10(radius(m, t)); };
        return solve_secant(f, m1, m1 * 2, 1e-10);
    }
    double density_at(double r, double t) const
    {
        const double r1 = envelop_outer_boundary(t);
        if (r < cloud_outer_boundary(t))
        {
            const double r_outer = cloud_outer_boundary(t);
            const double d_outer = density_at(r_outer, t);
            return d_outer * std::pow(r / r_outer, -cloud_index);
        }
        if (r > r1) return density_at(r1, t) * std::pow(r / r1, -2.0);
        return density(mass_coordinate(r, t), t);
    }
    double gamma_beta_at(double r, double t) const
    {
        const double r1 = envelop_outer_boundary(t);
        if (r < cloud_outer_boundary(t)) return cloud_gamma_beta();
        if (r > r1) return gamma_beta(mass_coordinate(r1, t));
        return gamma_beta(mass_coordinate(r, t));
    }
};

} // namespace model
