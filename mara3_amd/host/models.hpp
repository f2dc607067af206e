// Start-up and boundary data of the `cloud` sub-program, host side: the rest-mass density and four-velocity of (a) the engine's
// outflow through the inner boundary, (b) a broken power-law halo that only fixes the unit of mass, and (c) homologously expanding
// merger ejecta with a slow inner cloud. Evaluated once per radial zone at start-up and once per polar zone per step - never per
// cell per stage - and dominated by pow / exp / log10, so they run on the host with the host libm, like upstream.
//
// Layout: one plain parameter block per component, free functions over it. What is kept from upstream is the ORDER OF THE
// FLOATING-POINT OPERATIONS, because the sub-program's initial state and nozzle row must come out bit-identical to the reference's
// (tests/test_gpu_host_subprograms.py::test_cloud_subprogram_matches_reference compares them with reference-generated fixtures):
//   engine outflow      src/model_jet_nozzle.hpp:97-120      (four-velocity; base density)
//   halo                src/model_atmosphere.hpp:97-152      (profile and its two mass integrals)
//   ejecta              src/model_atmosphere.hpp:155-281     (mass-labelled shells; root find :56-78)
#pragma once
#include <cmath>
#include <stdexcept>

namespace model {

// ---- (a) engine outflow through the inner boundary ---------------------------------------------------------------------------
struct EngineParams
{
    double energy = 1.0;            // erg, both jets
    double four_velocity0 = 2.0;    // on the axis at t = 0
    double duration = 1.0;          // s
    double opening_angle = 0.1;     // rad
    double angular_exponent = 2.0;
    double base_radius = 1.0;       // cm
    double c_cgs = 3e10;            // upstream's engine model rounds the speed of light
};

// comoving density at the base for which the two jets carry `energy` in total (cold, ultra-relativistic estimate)
inline double engine_base_density(const EngineParams& e)
{
    const double area_factor = std::pow(e.four_velocity0 * e.base_radius * e.opening_angle, 2);
    return e.energy / (2 * M_PI * area_factor * e.duration * std::pow(e.c_cgs, 3));
}

// gamma-beta of one jet at polar angle `angle` and time `t`: Gaussian-like in angle, exponential in time
inline double engine_four_velocity(const EngineParams& e, double angle, double t)
{
    const double angular = std::exp(-0.5 * std::pow(angle / e.opening_angle, e.angular_exponent));
    const double temporal = std::exp(-0.5 * t / e.duration);
    return e.four_velocity0 * angular * temporal;
}

// ---- (b) broken power-law halo --------------------------------------------------------------------------------------------------
struct HaloParams
{
    double scale = 1.0;             // density at reference_radius
    double reference_radius = 1.0;
    double break_radius = 1e2;
    double inner_slope = 2.0, outer_slope = 6.0;
};

inline double halo_density(const HaloParams& h, double r)
{
    if (r <= h.break_radius) return h.scale * std::pow(r / h.reference_radius, -h.inner_slope);
    const double at_break = h.scale * std::pow(h.break_radius / h.reference_radius, -h.inner_slope);
    return at_break * std::pow(r / h.break_radius, -h.outer_slope);
}

inline double halo_mass(const HaloParams& h)
{
    if (h.outer_slope <= 3.0) throw std::invalid_argument("halo: the mass beyond the break diverges unless the outer slope exceeds 3");
    const double rb3 = std::pow(h.break_radius, 3), db = halo_density(h, h.break_radius);
    const double inside = h.inner_slope == 3.0
        ? 4 * M_PI * (db * rb3 * std::log(h.break_radius / h.reference_radius))
        : 4 * M_PI * (db * rb3 - halo_density(h, h.reference_radius) * std::pow(h.reference_radius, 3)) / (3 - h.inner_slope);
    const double outside = 4 * M_PI * db * rb3 / (h.outer_slope - 3);
    return inside + outside;
}

inline HaloParams halo_scaled_to_mass(HaloParams h, double mass)
{
    h.scale = mass / halo_mass(h);
    return h;
}

// ---- (c) homologous ejecta: shells labelled by the mass m outside them ----------------------------------------------------------
struct EjectaParams
{
    double shell_mass = 0.005 * 1.989e33;   // g: mass label of the slowest shell = where the inner cloud begins
    double fast_four_velocity = 4.0;        // gamma-beta of the shell labelled fast_mass
    double fast_mass = 1e26;                // g
    double velocity_slope = 0.25;           // gamma-beta ~ m^-slope
    double cloud_slope = 2.0;               // density ~ r^-slope inside the slowest shell
    double c_cgs = 2.998e10;
};

inline double shell_four_velocity(const EjectaParams& p, double m) { return p.fast_four_velocity * std::pow(m / p.fast_mass, -p.velocity_slope); }

inline double shell_speed(const EjectaParams& p, double m)
{
    const double u = shell_four_velocity(p, m);
    return u / std::sqrt(1.0 + u * u) * p.c_cgs;
}

inline double shell_radius(const EjectaParams& p, double m, double t) { return shell_speed(p, m) * t; }

inline double shell_density(const EjectaParams& p, double m, double t)
{
    const double lorentz2 = 1.0 + std::pow(shell_four_velocity(p, m), 2);
    const double beta = shell_speed(p, m) / p.c_cgs;
    const double du_dm = -p.velocity_slope / m * shell_four_velocity(p, m);
    return lorentz2 * beta / (4 * M_PI * std::pow(shell_radius(p, m, t), 3)) / std::abs(du_dm);
}

// secant iteration on `residual`, started from (a, b); returns the last iterate once |residual| <= tol
template<typename Residual>
double root_by_secant(Residual&& residual, double a, double b, double tol)
{
    double fa = residual(a), fb = residual(b);
    while (std::abs(fb) > tol)
    {
        const double c = b - fb * (b - a) / (fb - fa);
        a = b;
        fa = fb;
        b = c;
        fb = residual(c);
    }
    return b;
}

// mass label of the shell that is at radius r at time t (root of the log-radius mismatch)
inline double shell_label_at(const EjectaParams& p, double r, double t)
{
    return root_by_secant([&] (double m) { return std::log10(r) - std::log10(shell_radius(p, m, t)); }, p.fast_mass, p.fast_mass * 2, 1e-10);
}

struct EjectaState { double density, four_velocity; };

// rest-mass density and gamma-beta at radius r (cm) and time t (s): inner cloud | shells | r^-2 wind beyond the fastest shell
inline EjectaState ejecta_at(const EjectaParams& p, double r, double t)
{
    const double r_slow = shell_speed(p, p.shell_mass) * t;
    const double r_fast = shell_radius(p, p.fast_mass, t);
    if (r < r_slow)
    {
        const double beta = shell_speed(p, p.shell_mass) / p.c_cgs;
        const double at_edge = shell_density(p, shell_label_at(p, r_slow, t), t);
        return {at_edge * std::pow(r / r_slow, -p.cloud_slope), beta / std::sqrt(1.0 - beta * beta)};
    }
    if (r > r_fast)
    {
        const double m_fast = shell_label_at(p, r_fast, t);
        return {shell_density(p, m_fast, t) * std::pow(r / r_fast, -2.0), shell_four_velocity(p, m_fast)};
    }
    const double m = shell_label_at(p, r, t);
    return {shell_density(p, m, t), shell_four_velocity(p, m)};
}

} // namespace model
