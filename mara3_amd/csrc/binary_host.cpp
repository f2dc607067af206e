// Host-side scalar bookkeeping of the `binary` sub-program's solution_t: what binary::advance_u does with the
// source-term totals of one stage (src/subprog_binary_scheme.cpp:832-902) and the scalar part of the Runge-Kutta
// combine s0 * 1/2 + s2 * 1/2 (:1033-1069). A few dozen flops per stage, kept on the host exactly as upstream;
// compiled with g++ (like twobody.cpp) so that libm calls inside the orbital-element fit are scheduled as in the
// reference build.
#include <cmath>
#include <vector>
#include "../../include/mara_hip.h"
#include "binary_host.hpp"

namespace mh {

static mh_full_orbital_elements add(const mh_full_orbital_elements& a, const mh_full_orbital_elements& b)
{
    mh_full_orbital_elements r;
    r.pomega = a.pomega + b.pomega;
    r.tau = a.tau + b.tau;
    r.cm_position_x = a.cm_position_x + b.cm_position_x;
    r.cm_position_y = a.cm_position_y + b.cm_position_y;
    r.cm_velocity_x = a.cm_velocity_x + b.cm_velocity_x;
    r.cm_velocity_y = a.cm_velocity_y + b.cm_velocity_y;
    r.elements.separation = a.elements.separation + b.elements.separation;
    r.elements.total_mass = a.elements.total_mass + b.elements.total_mass;
    r.elements.mass_ratio = a.elements.mass_ratio + b.elements.mass_ratio;
    r.elements.eccentricity = a.elements.eccentricity + b.elements.eccentricity;
    return r;
}

static mh_full_orbital_elements scale(const mh_full_orbital_elements& a, double s)
{
    mh_full_orbital_elements r;
    r.pomega = a.pomega * s;
    r.tau = a.tau * s;
    r.cm_position_x = a.cm_position_x * s;
    r.cm_position_y = a.cm_position_y * s;
    r.cm_velocity_x = a.cm_velocity_x * s;
    r.cm_velocity_y = a.cm_velocity_y * s;
    r.elements.separation = a.elements.separation * s;
    r.elements.total_mass = a.elements.total_mass * s;
    r.elements.mass_ratio = a.elements.mass_ratio * s;
    r.elements.eccentricity = a.elements.eccentricity * s;
    return r;
}

// scheme.cpp:832-902: the state after one stage, given the stage's totals and the bodies it was evaluated with
int binary_apply_totals(const mh_binary_state& S, const mh_two_body_t& B, const double tot[MH_BINARY_NTOTALS], double dt,
                        bool no_accretion_force, double begin_live_binary, mh_binary_state* out)
{
    const double* b1 = B.body1;
    const double* b2 = B.body2;
    const double M1 = b1[0], M2 = b2[0];
    const double px1 = M1 * b1[3], py1 = M1 * b1[4], px2 = M2 * b2[3], py2 = M2 * b2[4];
    const double dM1 = tot[MH_T_MASS_ACC], dM2 = tot[MH_T_MASS_ACC + 1];
    const double vx1 = (px1 + tot[MH_T_PX_ACC]) / (M1 + dM1), vy1 = (py1 + tot[MH_T_PY_ACC]) / (M1 + dM1);
    const double vx2 = (px2 + tot[MH_T_PX_ACC + 1]) / (M2 + dM2), vy2 = (py2 + tot[MH_T_PY_ACC + 1]) / (M2 + dM2);
    mh_two_body_t acc, grv;
    acc.body1[0] = M1 + dM1; acc.body1[1] = b1[1]; acc.body1[2] = b1[2];
    acc.body1[3] = no_accretion_force ? b1[3] : vx1; acc.body1[4] = no_accretion_force ? b1[4] : vy1;
    acc.body2[0] = M2 + dM2; acc.body2[1] = b2[1]; acc.body2[2] = b2[2];
    acc.body2[3] = no_accretion_force ? b2[3] : vx2; acc.body2[4] = no_accretion_force ? b2[4] : vy2;
    grv.body1[0] = M1; grv.body1[1] = b1[1]; grv.body1[2] = b1[2];
    grv.body1[3] = b1[3] + tot[MH_T_FX] / M1; grv.body1[4] = b1[4] + tot[MH_T_FY] / M1;
    grv.body2[0] = M2; grv.body2[1] = b2[1]; grv.body2[2] = b2[2];
    grv.body2[3] = b2[3] + tot[MH_T_FX + 1] / M2; grv.body2[4] = b2[4] + tot[MH_T_FY + 1] / M2;

    const bool live = S.time > begin_live_binary;
    const mh_full_orbital_elements E0 = S.orbital_elements;
    mh_full_orbital_elements Ea, Eg, da, dg, dcm = {};
    if (int rc = mh_orbital_elements_from_state(&acc, S.time, &Ea)) return rc;
    if (int rc = mh_orbital_elements_from_state(&grv, S.time, &Eg)) return rc;
    mh_orbital_elements_diff(&E0, &Ea, &da);
    mh_orbital_elements_diff(&E0, &Eg, &dg);
    dcm.cm_position_x = E0.cm_velocity_x * dt;      // mara::diff_cm :520-526
    dcm.cm_position_y = E0.cm_velocity_y * dt;

    mh_binary_state R = S;
    R.time = S.time + dt;
    R.iteration = S.iteration + 1;
    for (int b = 0; b < 2; ++b)
    {
        R.mass_accreted_on[b] = S.mass_accreted_on[b] + tot[MH_T_MASS_ACC + b];
        R.angular_momentum_accreted_on[b] = S.angular_momentum_accreted_on[b] + tot[MH_T_L_ACC + b];
        R.integrated_torque_on[b] = S.integrated_torque_on[b] + tot[MH_T_TORQUE + b];
        R.work_done_on[b] = S.work_done_on[b] + tot[MH_T_WORK + b];
    }
    R.mass_ejected = S.mass_ejected + tot[MH_T_MASS_EJ];
    R.angular_momentum_ejected = S.angular_momentum_ejected + tot[MH_T_L_EJ];
    R.orbital_elements_acc = add(S.orbital_elements_acc, da);
    R.orbital_elements_grav = add(S.orbital_elements_grav, dg);
    R.orbital_elements = add(S.orbital_elements, scale(add(add(da, dg), dcm), live ? 1.0 : 0.0));
    *out = R;
    return MH_OK;
}

// the scalar part of s0 * b0 + s2 * (1 - b0), b0 = 1/2 (:1033-1069; subprog_binary.cpp:272-275). The iteration is a
// rational there: i / 2 + (i + 2) / 2 = i + 1 exactly.
void binary_combine_scalars(const mh_binary_state& a, const mh_binary_state& b, mh_binary_state* out)
{
    mh_binary_state r;
    r.time = a.time * 0.5 + b.time * 0.5;
    r.iteration = (a.iteration + b.iteration) / 2;
    for (int k = 0; k < 2; ++k)
    {
        r.mass_accreted_on[k] = a.mass_accreted_on[k] * 0.5 + b.mass_accreted_on[k] * 0.5;
        r.angular_momentum_accreted_on[k] = a.angular_momentum_accreted_on[k] * 0.5 + b.angular_momentum_accreted_on[k] * 0.5;
        r.integrated_torque_on[k] = a.integrated_torque_on[k] * 0.5 + b.integrated_torque_on[k] * 0.5;
        r.work_done_on[k] = a.work_done_on[k] * 0.5 + b.work_done_on[k] * 0.5;
    }
    r.mass_ejected = a.mass_ejected * 0.5 + b.mass_ejected * 0.5;
    r.angular_momentum_ejected = a.angular_momentum_ejected * 0.5 + b.angular_momentum_ejected * 0.5;
    r.orbital_elements_acc = add(scale(a.orbital_elements_acc, 0.5), scale(b.orbital_elements_acc, 0.5));
    r.orbital_elements_grav = add(scale(a.orbital_elements_grav, 0.5), scale(b.orbital_elements_grav, 0.5));
    r.orbital_elements = add(scale(a.orbital_elements, 0.5), scale(b.orbital_elements, 0.5));
    *out = r;
}

} // namespace mh

extern "C" {

int mh_binary_vertices(int block_size, int depth, double domain_radius, double* out)
{
    if (block_size < 1 || depth < 0 || depth > 20 || ! out) return MH_E_INVALID;
    int n = block_size;
    std::vector<double> a(n + 1), b;
    for (int i = 0; i <= n; ++i)
        a[i] = -1.0 + (1.0 - -1.0) * i / double((n + 1) - 1);          // nd::linspace core_ndarray.hpp:2544-2551
    for (int l = 0; l < depth; ++l)
    {
        b.resize(2 * n + 1);
        for (int i = 0; i <= 2 * n; ++i)
            b[i] = (a[i / 2] + a[(i + 1) / 2]) * 0.5;                    // prolong_verts mesh_prolong_restrict.hpp:148-159
        n *= 2;
        a.swap(b);
    }
    for (int i = 0; i <= n; ++i) out[i] = a[i] * domain_radius;
    return MH_OK;
}

int mh_binary_solver_data(const mh_binary_model* m, int n, const double* xv, const double* yv, double* u_init, double* br, double* recommended_dt)
{
    if (! m || n < 1 || ! xv || ! yv || ! u_init || ! br || ! recommended_dt) return MH_E_INVALID;
    const double rs = m->softening_radius, rc = m->disk_radius, Ma = m->mach_number;
    const double s0 = m->disk_mass / (17.0618 * rc * rc);
    const double s1 = m->ambient_density * s0;
    auto sigma = [=] (double r) { const double x = r / rc; return s0 * std::exp(-0.5 * (x - 1) * (x - 1)) + s1; };
    auto dp_dr = [=] (double r) { const double x = r / rc; return (1.0 / Ma / Ma / (r + rs)) * (x * (1 - x) * (1 - s1 / sigma(r)) - 1.0); };
    double min_dx = xv[1] - xv[0], min_dy = yv[1] - yv[0], max_v = 1.0;
    for (int i = 0; i < n; ++i)
    {
        min_dx = std::fmin(min_dx, xv[i + 1] - xv[i]);
        min_dy = std::fmin(min_dy, yv[i + 1] - yv[i]);
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
        {
            const double x = (xv[i] + xv[i + 1]) * 0.5, y = (yv[j] + yv[j + 1]) * 0.5;
            const double r2 = x * x + y * y;
            const double r = std::sqrt(r2);
            const double vp = std::sqrt(1.0 / (r + rs) + dp_dr(r)) * (m->counter_rotate ? -1 : 1);
            const double vr = -m->mdot / (sigma(r) * 2 * M_PI * r) * (r > 2.0);
            const double vx = vr * (x / r) + vp * (-y / r);
            const double vy = vr * (y / r) + vp * ( x / r);
            const double sg = sigma(r);
            double* u = u_init + 3 * ((size_t) i * n + j);
            u[0] = sg;
            if (m->angmom_form)
            {
                u[1] = sg * (x * vx + y * vy);                            // to_conserved_angmom_per_area physics_iso2d.hpp:263-272
                u[2] = sg * (x * vy - y * vx);
            }
            else
            {
                u[1] = sg * vx;                                           // to_conserved_per_area physics_iso2d.hpp:249-258
                u[2] = sg * vy;
            }
            const double v = std::sqrt(vx * vx + vy * vy);
            if (max_v < v) max_v = v;
            const double rcen = std::pow(x * x + y * y, 0.5);
            br[(size_t) i * n + j] = m->buffer_damping_rate * (1.0 + std::tanh(3.0 * (rcen - m->domain_radius)));
        }
    *recommended_dt = std::fmin(min_dx, min_dy) / max_v * m->cfl_number;
    return MH_OK;
}

} // extern "C"
