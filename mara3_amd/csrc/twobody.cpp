// Host-side Kepler two-body model: a handful of scalars per RK stage, kept on the host
// exactly as upstream (the device kernels receive the body positions and masses as arguments). Formulas and
// evaluation order follow
//   mara::compute_two_body_state(orbital_elements_t, t)       src/model_two_body.hpp:168-208
//   mara::compute_two_body_state(full_orbital_elements_t, t)  src/model_two_body.hpp:209-268
//   mara::compute_orbital_elements(two_body_state_t, t)       src/model_two_body.hpp:295-381
//   mara::orbital_period                                      src/model_two_body.hpp:391-396
//   mara::diff(full_orbital_elements_t, full_orbital_elements_t) src/model_two_body.hpp:492-518
// so that results are bit-identical to the reference (same libm: sin/cos/atan2/sqrt; no FMA contraction).
// No device code in this file: it is compiled with g++ like the reference (a host compiler may pair sin/cos calls
// or schedule libm differently; two of 2048 fixture rows differed by one ulp when clang compiled it).
#include <cmath>
#include <algorithm>
#include "../../include/mara_hip.h"

namespace mh { void set_error(const char* fmt, ...); }

namespace {

double newton(double e, double M)          // solve E - e sin E = M from E = M, |f| <= 1e-10  (:119-133, :194-196)
{
    double x = M;
    double y = x - e * std::sin(x) - M;
    while (std::abs(y) > 1e-10)
    {
        x -= y / (1 - e * std::cos(x));
        y = x - e * std::sin(x) - M;
    }
    return x;
}

void local_state(const mh_orbital_elements& p, double t, mh_two_body_t* s)
{
    const double e = p.eccentricity, q = p.mass_ratio, a = p.separation;
    const double omega = a == 0.0 ? 0.0 : std::sqrt(p.total_mass / a / a / a);
    const double mu = q / (1.0 + q);
    const double E = p.eccentricity > 0.0 ? newton(e, omega * t) : omega * t;
    s->body1[0] = p.total_mass * (1 - mu);
    s->body2[0] = p.total_mass * mu;
    s->body1[1] = -a * mu * (e - std::cos(E));
    s->body1[2] = +a * mu * (0 + std::sin(E)) * std::sqrt(1 - e * e);
    s->body2[1] = -s->body1[1] / q;
    s->body2[2] = -s->body1[2] / q;
    s->body1[3] = -a * mu * omega / (1 - e * std::cos(E)) * std::sin(E);
    s->body1[4] = +a * mu * omega / (1 - e * std::cos(E)) * std::cos(E) * std::sqrt(1 - e * e);
    s->body2[3] = -s->body1[3] / q;
    s->body2[4] = -s->body1[4] / q;
}

double period(const mh_orbital_elements& el)
{
    const double M = el.total_mass, a = el.separation;
    return 2 * M_PI / std::sqrt(M / a / a / a);
}

double clamp(double x0, double x1, double x) { return std::min(std::max(x, x0), x1); }

} // namespace

extern "C" {

int mh_two_body_state(const mh_full_orbital_elements* p, double t, mh_two_body_t* out)
{
    if (! p || ! out) return MH_E_INVALID;
    while (t < p->tau) t += period(p->elements);
    mh_two_body_t loc;
    local_state(p->elements, t - p->tau, &loc);
    const double c = std::cos(-p->pomega), s = std::sin(-p->pomega);
    const double* b[2] = {loc.body1, loc.body2};
    double* o[2] = {out->body1, out->body2};
    for (int k = 0; k < 2; ++k)
    {
        const double x = b[k][1], y = b[k][2], vx = b[k][3], vy = b[k][4];
        o[k][0] = b[k][0];
        o[k][1] = (+x * c + y * s) + p->cm_position_x;
        o[k][2] = (-x * s + y * c) + p->cm_position_y;
        o[k][3] = (+vx * c + vy * s) + p->cm_velocity_x;
        o[k][4] = (-vx * s + vy * c) + p->cm_velocity_y;
    }
    return MH_OK;
}

int mh_orbital_elements_from_state(const mh_two_body_t* st, double t, mh_full_orbital_elements* P)
{
    if (! st || ! P) return MH_E_INVALID;
    const double* c1 = st->body1;
    const double* c2 = st->body2;
    const double M1 = c1[0], M2 = c2[0];
    const double M = M1 + M2;
    const double q = M2 / M1;
    const double x_cm  = (c1[1] * c1[0] + c2[1] * c2[0]) / M;
    const double y_cm  = (c1[2] * c1[0] + c2[2] * c2[0]) / M;
    const double vx_cm = (c1[3] * c1[0] + c2[3] * c2[0]) / M;
    const double vy_cm = (c1[4] * c1[0] + c2[4] * c2[0]) / M;
    const double x1 = c1[1] - x_cm, y1 = c1[2] - y_cm, x2 = c2[1] - x_cm, y2 = c2[2] - y_cm;
    const double r1 = std::sqrt(x1 * x1 + y1 * y1);
    const double r2 = std::sqrt(x2 * x2 + y2 * y2);
    const double vx1 = c1[3] - vx_cm, vy1 = c1[4] - vy_cm, vx2 = c2[3] - vx_cm, vy2 = c2[4] - vy_cm;
    const double vf1 = -vx1 * y1 / r1 + vy1 * x1 / r1;
    const double vf2 = -vx2 * y2 / r2 + vy2 * x2 / r2;
    const double v1 = std::sqrt(vx1 * vx1 + vy1 * vy1);
    const double E1 = 0.5 * M1 * (vx1 * vx1 + vy1 * vy1);
    const double E2 = 0.5 * M2 * (vx2 * vx2 + vy2 * vy2);
    const double L1 = M1 * r1 * vf1;
    const double L2 = M2 * r2 * vf2;
    const double R = r1 + r2;
    const double L = L1 + L2;
    const double E = E1 + E2 - M1 * M2 / R;
    const double a = -0.5 * M1 * M2 / E;
    const double b = std::sqrt(-0.5 * L * L / E * (M1 + M2) / (M1 * M2));
    const double e = std::sqrt(clamp(0.0, 1.0, 1.0 - b * b / a / a));
    const double omega = std::sqrt(M / a / a / a);
    const double a1 = a * q / (1.0 + q);
    const double b1 = b * q / (1.0 + q);
    const double cn = e == 0.0 ? x1 / r1 : (1.0 - r1 / a1) / e;
    const double cf = a1 / r1 * (cn - e);
    const double sn = e == 0.0 ? y1 / r1 : (vx1 * x1 + vy1 * y1) / (e * v1 * r1) * std::sqrt(1.0 - e * e * cn * cn);
    const double sf = (b1 / r1) * sn;
    const double cE = (e + cf) / (1.0 + e * cf);
    const double sE = std::sqrt(1.0 - e * e) * sf / (1.0 + e * cf);
    const double EE = std::atan2(sE, cE);
    const double MM = EE - e * sE;
    const double tau = t - MM / omega;
    const double ax = +(cn - e) * x1 + sn * std::sqrt(1.0 - e * e) * y1;
    const double ay = +(cn - e) * y1 - sn * std::sqrt(1.0 - e * e) * x1;
    const double pomega = std::atan2(ay, ax);
    if (E >= 0.0)
    {
        mh::set_error("mara::compute_orbital_elements (two_body_state does not correspond to a bound orbit)");
        return MH_E_PHYSICS;
    }
    P->tau = tau;
    P->pomega = pomega;
    P->cm_position_x = x_cm;
    P->cm_position_y = y_cm;
    P->cm_velocity_x = vx_cm;
    P->cm_velocity_y = vy_cm;
    P->elements.separation = a;
    P->elements.total_mass = M;
    P->elements.mass_ratio = q;
    P->elements.eccentricity = e;
    return MH_OK;
}

void mh_orbital_elements_diff(const mh_full_orbital_elements* a, const mh_full_orbital_elements* b, mh_full_orbital_elements* out)
{
    // mara::diff :492-518
    auto wrap = [] (double delta, double per)
    {
        const double x = delta, y = delta + per, z = delta - per;
        if (std::abs(x) < std::min(std::abs(y), std::abs(z))) return x;
        if (std::abs(y) < std::abs(z)) return y;
        return z;
    };
    mh_full_orbital_elements r;
    r.pomega = wrap(b->pomega - a->pomega, 2 * M_PI);
    r.tau = wrap(b->tau - a->tau, period(b->elements));
    r.cm_position_x = b->cm_position_x - a->cm_position_x;
    r.cm_position_y = b->cm_position_y - a->cm_position_y;
    r.cm_velocity_x = b->cm_velocity_x - a->cm_velocity_x;
    r.cm_velocity_y = b->cm_velocity_y - a->cm_velocity_y;
    r.elements.separation = b->elements.separation - a->elements.separation;
    r.elements.total_mass = b->elements.total_mass - a->elements.total_mass;
    r.elements.mass_ratio = b->elements.mass_ratio - a->elements.mass_ratio;
    r.elements.eccentricity = b->elements.eccentricity - a->elements.eccentricity;
    *out = r;
}

} // extern "C"
