/**
 * TEST INFRASTRUCTURE — reference-side oracle driver for BASELINE config 3
 * (`mara binary`) on a uniform-depth
 * block tree (focus_factor=1e9: every node refines).
 *
 * The sub-program's translation units cannot be compiled here (they include a
 * generated header, app_compile_opts.hpp, and core_hdf5.hpp), so — like
 * sedov_ref / cloud_ref — this driver composes the reference HEADER functions in
 * the order the sub-program applies them:
 *
 *   mara::create_vertex_quadtree             mesh_tree_operators.hpp:158  (the real block tree -> vertices)
 *   mara::iso2d::recover_primitive           physics_iso2d.hpp:351
 *   mara::plm_gradient(primitive_t...)       math_interpolation.hpp:85-133
 *   mara::iso2d::riemann_hlle                physics_iso2d.hpp:488
 *   primitive_t::to_conserved_per_area, max_wavespeed, iso2d::angular_momentum
 *   mara::compute_two_body_state / compute_orbital_elements / diff / diff_cm   model_two_body.hpp
 *
 * What is NOT a header function — the arithmetic written inline in the scheme's
 * translation unit (softened gravity, sink, buffer, sound speed, alpha viscosity, the
 * update expression: subprog_binary_scheme.cpp:85-126,160-193,220-293,345-411,568-587,
 * 790-904; IC and dt: subprog_binary.cpp:105-153,258-293, solver_data.cpp:45-102) —
 * is restated below in plain doubles with the reference's operation order. Those
 * lines are a restatement, so the scheme-level vectors this driver emits pin the
 * composition "reference leaf functions + restated glue", not the sub-program binary.
 *
 * usage: binary_ref <out_prefix> [key=value ...]
 *   keys: the sub-program's config names (depth, block_size, rk_order, fixed_dt, plm_theta, ...)
 *         plus nsteps=<int> (full RK steps to take)
 * writes <out_prefix>.{xv,yv,u_init,br,u_stage,stage_scalars,u_final,scalars}.f64
 *   stage_scalars = dt, recommended_time_step, maximum_timestep, acc[10], E_acc[10], E_grav[10], E[10] after ONE advance_u from
 *                   the initial state, its 18 source-term totals, and the two bodies (mass, x, y, vx, vy) at the initial time
 *   scalars       = time, iteration, acc[10], E_acc[10], E_grav[10], E[10] after nsteps full RK steps, then the dt of each step
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <numeric>
#include <functional>
#include <map>
#include <string>
#include <vector>
#include <stdexcept>
#include "core_ndarray.hpp"
#include "core_ndarray_ops.hpp"
#include "core_dimensional.hpp"
#include "core_sequence.hpp"
#include "core_tuple.hpp"
#include "core_geometric.hpp"
#include "core_rational.hpp"
#include "core_tree.hpp"
#include "math_interpolation.hpp"
#include "mesh_prolong_restrict.hpp"
#include "mesh_tree_operators.hpp"
#include "model_two_body.hpp"
#include "physics_iso2d.hpp"

using prim_t = mara::iso2d::primitive_t;
using cons_t = mara::iso2d::conserved_per_area_t;
using loc_t  = mara::iso2d::location_2d_t;

struct cons3 { double s, px, py; };
static cons_t to_ref(cons3 u)
{
    return cons_t().set<0>(mara::make_dimensional<-2, 1, 0>(u.s)).set<1>(mara::make_dimensional<-1, 1, -1>(u.px)).set<2>(mara::make_dimensional<-1, 1, -1>(u.py));
}
static cons3 from_ref(const cons_t& U) { return {mara::get<0>(U).value, mara::get<1>(U).value, mara::get<2>(U).value}; }
static loc_t make_loc(double x, double y) { return {{mara::make_length(x), mara::make_length(y)}}; }
using consq_t = mara::iso2d::conserved_angmom_per_area_t;
static consq_t to_ref_q(cons3 q)
{
    return consq_t().set<0>(mara::make_dimensional<-2, 1, 0>(q.s)).set<1>(mara::make_dimensional<0, 1, -1>(q.px)).set<2>(mara::make_dimensional<0, 1, -1>(q.py));
}
static cons3 from_ref_q(const consq_t& Q) { return {mara::get<0>(Q).value, mara::get<1>(Q).value, mara::get<2>(Q).value}; }

enum { T_MASS_ACC = 0, T_L_ACC = 2, T_TORQUE = 4, T_PX_ACC = 6, T_PY_ACC = 8, T_FX = 10, T_FY = 12, T_WORK = 14, T_MASS_EJ = 16, T_L_EJ = 17, NTOT = 18 };

struct params_t
{
    std::map<std::string, double> cfg;
    double get(const char* k) const { return cfg.at(k); }
    int N = 0, bs = 0, depth = 0;
    std::vector<double> xv, yv, br;
    std::vector<cons3> u_init;
    double h = 0, gst = 0, recommended_dt = 0;
    bool qform = false;      // conserve_linear_p == 0: the cons3 fields hold (Sigma, Sigma s_r, Sigma l_z), scheme.cpp:906-1020
};

struct solution_t
{
    double time = 0;
    mara::rational_number_t iteration = 0;
    std::vector<cons3> u;
    double acc[10] = {0}; // mass_acc[2], L_acc[2], torque[2], work[2], mass_ej, L_ej
    mara::full_orbital_elements_t E_acc, E_grav, E;
};

static prim_t disk_profile(const params_t& P, double x, double y)
{
    // subprog_binary.cpp:105-153
    double rs = P.get("softening_radius"), rc = P.get("disk_radius"), Ma = P.get("mach_number");
    double s0 = P.get("disk_mass") / (17.0618 * rc * rc);
    double s1 = P.get("ambient_density") * s0;
    double mdot = P.get("mdot");
    auto sigma = [=] (double r) { auto q = r / rc; return s0 * std::exp(-0.5 * (q - 1) * (q - 1)) + s1; };
    auto dp_dr = [=] (double r) { auto q = r / rc; return (1.0 / Ma / Ma / (r + rs)) * (q * (1 - q) * (1 - s1 / sigma(r)) - 1.0); };
    double r2 = x * x + y * y;
    double r = std::sqrt(r2);
    double vp = std::sqrt(1.0 / (r + rs) + dp_dr(r)) * (int(P.get("counter_rotate")) ? -1 : 1);
    double vr = -mdot / (sigma(r) * 2 * M_PI * r) * (r > 2.0);
    double vx = vr * (x / r) + vp * (-y / r);
    double vy = vr * (y / r) + vp * ( x / r);
    return prim_t().with_sigma(sigma(r)).with_velocity_x(vx).with_velocity_y(vy);
}

static double phi_soft(const params_t& P, double x, double y, const mara::point_mass_t& b)
{
    // grav_phi_field, scheme.cpp:101-111
    double d0 = x - b.position_x, d1 = y - b.position_y;
    double dr2 = d0 * d0 + d1 * d1;
    double rs2 = P.get("softening_radius") * P.get("softening_radius");
    return -1.0 * b.mass / std::pow(dr2 + rs2, 0.5);
}
static double cs2_at(const params_t& P, double x, double y, const mara::two_body_state_t& B)
{
    // scheme.cpp:160-175
    double M = P.get("mach_number");
    if (int(P.get("axisymmetric_cs2")))
        return 1.0 / std::sqrt(x * x + y * y) / M / M;
    return -(phi_soft(P, x, y, B.body1) + phi_soft(P, x, y, B.body2)) / M / M;
}
static double nu_at(const params_t& P, double x, double y, double cs2)
{
    // scheme.cpp:177-193
    double radius = std::sqrt(x * x + y * y);
    double rc = P.get("alpha_cutoff_radius");
    double profile = rc > 0.0 ? 0.5 * (1.0 + std::tanh(3.0 * (radius - rc))) : 1.0;
    if (P.get("nu") > 0.0)
        return profile * P.get("nu");
    return profile * P.get("alpha") * std::sqrt(cs2) * (radius / P.get("mach_number"));
}

static cons3 face_flux(const params_t& P, int axis, double xf, double yf, const mara::two_body_state_t& B,
    prim_t pl, prim_t pr, prim_t gl, prim_t gr, prim_t hl, prim_t hr)
{
    // intercell_flux_u scheme.cpp:268-293 + viscous_flux :220-262
    auto pl_hat = pl + gl * 0.5 * P.h;
    auto pr_hat = pr - gr * 0.5 * P.h;
    double cs2 = cs2_at(P, xf, yf, B);
    double nu = nu_at(P, xf, yf, cs2);
    double mu = 0.5 * nu * (pl_hat.sigma() + pr_hat.sigma());
    auto F = mara::iso2d::riemann_hlle(pl_hat, pr_hat, cs2, cs2, mara::unit_vector_t::on_axis(std::size_t(axis)));
    double f0 = mara::get<0>(F).value, f1 = mara::get<1>(F).value, f2 = mara::get<2>(F).value;
    double v1, v2;
    if (axis == 0)
    {
        double dx_ux = 0.5 * (gl.velocity_x() + gr.velocity_x());
        double dx_uy = 0.5 * (gl.velocity_y() + gr.velocity_y());
        double dy_ux = 0.5 * (hl.velocity_x() + hr.velocity_x());
        double dy_uy = 0.5 * (hl.velocity_y() + hr.velocity_y());
        v1 = -(mu * (dx_ux - dy_uy));
        v2 = -(mu * (dx_uy + dy_ux));
    }
    else
    {
        double dx_ux = 0.5 * (hl.velocity_x() + hr.velocity_x());
        double dx_uy = 0.5 * (hl.velocity_y() + hr.velocity_y());
        double dy_ux = 0.5 * (gl.velocity_x() + gr.velocity_x());
        double dy_uy = 0.5 * (gl.velocity_y() + gr.velocity_y());
        v1 = -(mu * (dx_uy + dy_ux));
        v2 = -(-mu * (dx_ux - dy_uy));
    }
    cons3 f = {f0 + 0.0, f1 + v1, f2 + v2};
    if (P.qform)
    {
        // to_angmom_fluxes scheme.cpp:199-214
        double rd = P.get("domain_radius");
        double flux_sr = xf * f.px + yf * f.py;
        double flux_lz = xf * f.py - yf * f.px;
        if (axis == 0 && (xf == -rd || xf == rd)) flux_lz = 0.0;
        if (axis == 1 && (yf == -rd || yf == rd)) flux_lz = 0.0;
        f = {f.s, flux_sr, flux_lz};
    }
    return f;
}

static double fold_tree(const std::vector<double>& block_vals, int nb, int level, int depth, int bi, int bj)
{
    // arithmetic_binary_tree_t::sum (core_tree.hpp:502) over sequence_t::sum (core_sequence.hpp:216):
    // children in orthant order (i + 2 j), folded from zero.
    if (level == depth) return block_vals[bi * nb + bj];
    double r = 0.0;
    for (int c = 0; c < 4; ++c)
        r = r + fold_tree(block_vals, nb, level + 1, depth, bi * 2 + (c & 1), bj * 2 + ((c >> 1) & 1));
    return r;
}

static solution_t advance_u(const params_t& P, const solution_t& S, double dt, bool safe_mode, double* totals_out=nullptr)
{
    const int N = P.N, bs = P.bs, nb = N / bs;
    const double th = safe_mode ? 0.0 : P.get("plm_theta");
    auto at = [N] (int i, int j) { return std::size_t(((i + N) % N)) * N + ((j + N) % N); };
    std::vector<prim_t> p(std::size_t(N) * N), gx(p.size()), gy(p.size());
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
        {
            std::size_t n = at(i, j);
            p[n] = P.qform
            ? mara::iso2d::recover_primitive(to_ref_q(S.u[n]), make_loc((P.xv[i] + P.xv[i + 1]) * 0.5, (P.yv[j] + P.yv[j + 1]) * 0.5))
            : mara::iso2d::recover_primitive(to_ref(S.u[n]));
        }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
        {
            gx[at(i, j)] = mara::plm_gradient(p[at(i - 1, j)], p[at(i, j)], p[at(i + 1, j)], th) / P.h;
            gy[at(i, j)] = mara::plm_gradient(p[at(i, j - 1)], p[at(i, j)], p[at(i, j + 1)], th) / P.h;
        }
    auto B = mara::compute_two_body_state(S.E, S.time);
    const mara::point_mass_t bodies[2] = {B.body1, B.body2};

    // fluxes times transverse length, one per face *of each cell* (the positions of a block's outer
    // faces are its own vertices, so the two sides of the periodic seam are evaluated separately)
    std::vector<cons3> fx(std::size_t(N + 1) * N), fy(std::size_t(N) * (N + 1));
    for (int i = 0; i <= N; ++i)
        for (int j = 0; j < N; ++j)
        {
            double xf = (P.xv[i] + P.xv[i]) * 0.5, yf = (P.yv[j] + P.yv[j + 1]) * 0.5;
            auto f = face_flux(P, 0, xf, yf, B, p[at(i - 1, j)], p[at(i, j)], gx[at(i - 1, j)], gx[at(i, j)], gy[at(i - 1, j)], gy[at(i, j)]);
            double dy = P.yv[j + 1] - P.yv[j];
            fx[std::size_t(i) * N + j] = {f.s * dy, f.px * dy, f.py * dy};
        }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j <= N; ++j)
        {
            double xf = (P.xv[i] + P.xv[i + 1]) * 0.5, yf = (P.yv[j] + P.yv[j]) * 0.5;
            auto f = face_flux(P, 1, xf, yf, B, p[at(i, j - 1)], p[at(i, j)], gy[at(i, j - 1)], gy[at(i, j)], gx[at(i, j - 1)], gx[at(i, j)]);
            double dx = P.xv[i + 1] - P.xv[i];
            fy[std::size_t(i) * (N + 1) + j] = {f.s * dx, f.px * dx, f.py * dx};
        }

    solution_t R = S;
    R.u.resize(S.u.size());
    std::vector<double> blk[NTOT];
    for (auto& b : blk) b.assign(std::size_t(nb) * nb, 0.0);
    double rs2 = P.get("softening_radius") * P.get("softening_radius");
    double s2 = P.get("sink_radius") * P.get("sink_radius");
    double floor_sigma = P.get("density_floor") * P.get("disk_mass");
    bool negative = false;

    for (int bi = 0; bi < nb; ++bi)
        for (int bj = 0; bj < nb; ++bj)
        {
            double t[NTOT] = {0};
            double sink_sum[2][3] = {{0}};
            for (int i = bi * bs; i < (bi + 1) * bs; ++i)
                for (int j = bj * bs; j < (bj + 1) * bs; ++j)
                {
                    double xc = (P.xv[i] + P.xv[i + 1]) * 0.5, yc = (P.yv[j] + P.yv[j + 1]) * 0.5;
                    double dA = (P.xv[i + 1] - P.xv[i]) * (P.yv[j + 1] - P.yv[j]);
                    cons3 u0 = S.u[at(i, j)];
                    auto lz = [xc, yc] (cons3 u) { return mara::iso2d::angular_momentum(to_ref(u), make_loc(xc, yc)).value; };
                    cons3 s_grav[2], s_sink[2];
                    double fg[2][2];
                    for (int b = 0; b < 2; ++b)
                    {
                        // grav_vdot_field :85-95, times sigma (:369-370)
                        double d0 = xc - bodies[b].position_x, d1 = yc - bodies[b].position_y;
                        double den = std::pow(d0 * d0 + d1 * d1 + rs2, 1.5);
                        fg[b][0] = (-d0 / den * 1.0 * bodies[b].mass) * u0.s;
                        fg[b][1] = (-d1 / den * 1.0 * bodies[b].mass) * u0.s;
                        s_grav[b] = {0.0 * dt, fg[b][0] * dt, fg[b][1] * dt};
                        // sink_rate_field :117-126
                        double a2 = (d0 * d0 + d1 * d1) / s2 / 2.0;
                        double rate = P.get("sink_rate") * std::exp(-a2);
                        s_sink[b] = {-u0.s * rate * dt, -u0.px * rate * dt, -u0.py * rate * dt};
                    }
                    cons3 ui = P.u_init[at(i, j)];
                    double br = P.br[at(i, j)];
                    cons3 s_buffer = {(ui.s - u0.s) * br * dt, (ui.px - u0.px) * br * dt, (ui.py - u0.py) * br * dt};
                    double fl = double(u0.s < floor_sigma);
                    cons3 s_floor = {u0.s * 1e-2 * fl, u0.px * 1e-2 * fl, u0.py * 1e-2 * fl};
                    cons3 dps[2] = {s_sink[0], s_sink[1]};
                    if (P.qform)
                    {
                        // source_terms_q :417-466: gravity in (s_r, l_z) form, a geometrical source instead of the floor term
                        for (int b = 0; b < 2; ++b)
                        {
                            s_grav[b] = {0.0 * dt, (xc * fg[b][0] + yc * fg[b][1]) * dt, (xc * fg[b][1] - yc * fg[b][0]) * dt};
                            dps[b] = from_ref(mara::iso2d::to_conserved_per_area(to_ref_q(s_sink[b]), make_loc(xc, yc)));
                        }
                        double sr2 = std::pow(P.gst, 2.0);           // gst_suppr_radius.pow<2>() :421
                        double ramp = 1.0 - std::exp(-(xc * xc + yc * yc) / sr2);
                        auto sg = p[at(i, j)].source_terms_conserved_angmom(cs2_at(P, xc, yc, B));
                        s_floor = {mara::get<0>(sg).value * ramp * dt, mara::get<1>(sg).value * ramp * dt, mara::get<2>(sg).value * ramp * dt};
                    }

                    for (int b = 0; b < 2; ++b)
                    {
                        t[T_MASS_ACC + b] = t[T_MASS_ACC + b] + s_sink[b].s * dA;
                        t[T_L_ACC + b]    = t[T_L_ACC + b] + (P.qform ? s_sink[b].py : lz(s_sink[b])) * dA;
                        t[T_TORQUE + b]   = t[T_TORQUE + b] + (P.qform ? s_grav[b].py : lz(s_grav[b])) * dA;
                        t[T_FX + b]       = t[T_FX + b] + fg[b][0] * dt * dA;
                        t[T_FY + b]       = t[T_FY + b] + fg[b][1] * dt * dA;
                        t[T_PX_ACC + b]   = t[T_PX_ACC + b] + dps[b].px * dA;
                        t[T_PY_ACC + b]   = t[T_PY_ACC + b] + dps[b].py * dA;
                        sink_sum[b][0] = sink_sum[b][0] + s_sink[b].s * dA;
                        sink_sum[b][1] = sink_sum[b][1] + s_sink[b].px * dA;
                        sink_sum[b][2] = sink_sum[b][2] + s_sink[b].py * dA;
                    }
                    t[T_L_EJ]    = t[T_L_EJ] + (P.qform ? s_buffer.py : lz(s_buffer)) * dA;
                    t[T_MASS_EJ] = t[T_MASS_EJ] + s_buffer.s * dA;

                    // block_update_u :568-587
                    cons3 fxl = fx[std::size_t(i) * N + j], fxr = fx[std::size_t(i + 1) * N + j];
                    cons3 fyl = fy[std::size_t(i) * (N + 1) + j], fyr = fy[std::size_t(i) * (N + 1) + j + 1];
                    auto upd = [dt, dA] (double u, double xl, double xr, double yl, double yr, double g1, double g2, double k1, double k2, double bf, double flr)
                    {
                        double l = (xr - xl) + (yr - yl);
                        double s = g1 + g2 + k1 + k2 + bf + flr;
                        return u - l * dt / dA + s;
                    };
                    cons3 u1 = {
                        upd(u0.s,  fxl.s,  fxr.s,  fyl.s,  fyr.s,  s_grav[0].s,  s_grav[1].s,  s_sink[0].s,  s_sink[1].s,  s_buffer.s,  s_floor.s),
                        upd(u0.px, fxl.px, fxr.px, fyl.px, fyr.px, s_grav[0].px, s_grav[1].px, s_sink[0].px, s_sink[1].px, s_buffer.px, s_floor.px),
                        upd(u0.py, fxl.py, fxr.py, fyl.py, fyr.py, s_grav[0].py, s_grav[1].py, s_sink[0].py, s_sink[1].py, s_buffer.py, s_floor.py)};
                    if (u1.s < 0.0) negative = true;
                    R.u[at(i, j)] = u1;
                }
            for (int k = 0; k < NTOT; ++k) t[k] = -t[k];
            for (int b = 0; b < 2; ++b)
            {
                // work :356-365 with du = -(sum of s_sink dA)
                double M0 = bodies[b].mass, px0 = bodies[b].velocity_x * M0, py0 = bodies[b].velocity_y * M0;
                double M1 = M0 + -sink_sum[b][0], px1 = px0 + -sink_sum[b][1], py1 = py0 + -sink_sum[b][2];
                t[T_WORK + b] = ((px1 * px1 + py1 * py1) / M1 - (px0 * px0 + py0 * py0) / M0) * 0.5;
                if (P.qform) t[T_WORK + b] = 0.0;      // source_terms_q leaves work_done_on at its initial zero
            }
            for (int k = 0; k < NTOT; ++k) blk[k][bi * nb + bj] = t[k];
        }
    if (negative) throw std::runtime_error("negative density in updated state");

    double tot[NTOT];
    for (int k = 0; k < NTOT; ++k) tot[k] = fold_tree(blk[k], nb, 0, P.depth, 0, 0);

    // orbital-element perturbations, scheme.cpp:832-885
    double M1 = B.body1.mass, M2 = B.body2.mass;
    double px1 = M1 * B.body1.velocity_x, py1 = M1 * B.body1.velocity_y, px2 = M2 * B.body2.velocity_x, py2 = M2 * B.body2.velocity_y;
    double dM1 = tot[T_MASS_ACC], dM2 = tot[T_MASS_ACC + 1];
    double vx1 = (px1 + tot[T_PX_ACC]) / (M1 + dM1), vy1 = (py1 + tot[T_PY_ACC]) / (M1 + dM1);
    double vx2 = (px2 + tot[T_PX_ACC + 1]) / (M2 + dM2), vy2 = (py2 + tot[T_PY_ACC + 1]) / (M2 + dM2);
    bool naf = int(P.get("no_accretion_force"));
    mara::point_mass_t b1a = {M1 + dM1, B.body1.position_x, B.body1.position_y, naf ? B.body1.velocity_x : vx1, naf ? B.body1.velocity_y : vy1};
    mara::point_mass_t b2a = {M2 + dM2, B.body2.position_x, B.body2.position_y, naf ? B.body2.velocity_x : vx2, naf ? B.body2.velocity_y : vy2};
    mara::point_mass_t b1g = {M1, B.body1.position_x, B.body1.position_y, B.body1.velocity_x + tot[T_FX] / M1, B.body1.velocity_y + tot[T_FY] / M1};
    mara::point_mass_t b2g = {M2, B.body2.position_x, B.body2.position_y, B.body2.velocity_x + tot[T_FX + 1] / M2, B.body2.velocity_y + tot[T_FY + 1] / M2};
    bool live = S.time > P.get("begin_live_binary");
    auto E0 = S.E;
    auto Ea = mara::compute_orbital_elements({b1a, b2a}, S.time);
    auto Eg = mara::compute_orbital_elements({b1g, b2g}, S.time);

    R.time = S.time + dt;
    R.iteration = S.iteration + 1;
    const int acc_from[10] = {T_MASS_ACC, T_MASS_ACC + 1, T_L_ACC, T_L_ACC + 1, T_TORQUE, T_TORQUE + 1, T_WORK, T_WORK + 1, T_MASS_EJ, T_L_EJ};
    for (int k = 0; k < 10; ++k) R.acc[k] = S.acc[k] + tot[acc_from[k]];
    R.E_acc  = S.E_acc  + mara::diff(E0, Ea);
    R.E_grav = S.E_grav + mara::diff(E0, Eg);
    R.E      = S.E + (mara::diff(E0, Ea) + mara::diff(E0, Eg) + mara::diff_cm(E0, dt)) * double(live);
    if (totals_out) for (int k = 0; k < NTOT; ++k) totals_out[k] = tot[k];
    return R;
}

static solution_t combine(const solution_t& a, const solution_t& b)
{
    // s0 * 1/2 + s2 * 1/2 : scheme.cpp:1033-1069
    auto half = mara::make_rational(1, 2);
    solution_t r;
    r.time = a.time * half.as_double() + b.time * half.as_double();
    r.iteration = a.iteration * half + b.iteration * half;
    r.u.resize(a.u.size());
    for (std::size_t n = 0; n < a.u.size(); ++n)
        r.u[n] = {a.u[n].s * 0.5 + b.u[n].s * 0.5, a.u[n].px * 0.5 + b.u[n].px * 0.5, a.u[n].py * 0.5 + b.u[n].py * 0.5};
    for (int k = 0; k < 10; ++k) r.acc[k] = a.acc[k] * 0.5 + b.acc[k] * 0.5;
    r.E_acc = a.E_acc * 0.5 + b.E_acc * 0.5;
    r.E_grav = a.E_grav * 0.5 + b.E_grav * 0.5;
    r.E = a.E * 0.5 + b.E * 0.5;
    return r;
}

static double maximum_timestep(const params_t& P, const solution_t& S)
{
    // scheme.cpp:1107-1126 (uniform tree: one spacing)
    auto B = mara::compute_two_body_state(S.E, S.time);
    const int N = P.N, bs = P.bs, nb = N / bs;
    double result = 0.0; bool first = true;
    for (int bi = 0; bi < nb; ++bi)
        for (int bj = 0; bj < nb; ++bj)
        {
            double a = 0.0; bool f0 = true;
            for (int i = bi * bs; i < (bi + 1) * bs; ++i)
                for (int j = bj * bs; j < (bj + 1) * bs; ++j)
                {
                    double xc = (P.xv[i] + P.xv[i + 1]) * 0.5, yc = (P.yv[j] + P.yv[j + 1]) * 0.5;
                    auto p = P.qform ? mara::iso2d::recover_primitive(to_ref_q(S.u[std::size_t(i) * N + j]), make_loc(xc, yc))
                                     : mara::iso2d::recover_primitive(to_ref(S.u[std::size_t(i) * N + j]));
                    double w = p.max_wavespeed(cs2_at(P, xc, yc, B));
                    a = f0 ? w : std::max(a, w); f0 = false;
                }
            double d = P.h / a;
            result = first ? d : std::min(result, d); first = false;
        }
    return result;
}

static void dump(const std::string& name, const void* data, std::size_t bytes)
{
    FILE* f = std::fopen(name.data(), "wb");
    std::fwrite(data, 1, bytes, f);
    std::fclose(f);
}
static void push_elements(std::vector<double>& v, const mara::full_orbital_elements_t& P)
{
    for (double x : {P.pomega, P.tau, P.cm_position_x, P.cm_position_y, P.cm_velocity_x, P.cm_velocity_y,
                     P.elements.separation, P.elements.total_mass, P.elements.mass_ratio, P.elements.eccentricity}) v.push_back(x);
}

int main(int argc, char** argv)
{
    if (argc < 2) return 1;
    std::string prefix = argv[1];
    params_t P;
    P.cfg = { // subprog_binary.cpp:55-99
        {"cfl_number", 0.4}, {"fixed_dt", 0}, {"depth", 4}, {"begin_live_binary", 1e6}, {"block_size", 24}, {"rk_order", 2},
        {"plm_theta", 1.8}, {"source_term_softening", 1.}, {"softening_radius", 0.05}, {"sink_radius", 0.05}, {"sink_rate", 1.0},
        {"buffer_damping_rate", 10.0}, {"domain_radius", 12.0}, {"disk_radius", 2.0}, {"disk_mass", 1e-3}, {"ambient_density", 1e-4},
        {"density_floor", 0.0}, {"separation", 1.0}, {"mass_ratio", 1.0}, {"eccentricity", 0.0}, {"counter_rotate", 0},
        {"mach_number", 10.0}, {"axisymmetric_cs2", 0}, {"no_accretion_force", 0}, {"alpha_cutoff_radius", 0.0}, {"alpha", 0.1},
        {"nu", 0.0}, {"mdot", 0.0}, {"nsteps", 1}, {"safe_mode", 0}, {"conserve_linear_p", 1}};
    for (int a = 2; a < argc; ++a)
    {
        std::string kv = argv[a];
        auto eq = kv.find('=');
        if (eq == std::string::npos || ! P.cfg.count(kv.substr(0, eq))) { std::fprintf(stderr, "bad argument %s\n", argv[a]); return 1; }
        P.cfg[kv.substr(0, eq)] = std::atof(kv.substr(eq + 1).data());
    }
    P.qform = int(P.get("conserve_linear_p")) == 0;
    P.depth = int(P.get("depth"));
    P.bs = int(P.get("block_size"));
    P.N = P.bs << P.depth;
    const int N = P.N, bs = P.bs, nb = N / bs;
    const double R = P.get("domain_radius");

    // the real block tree of vertices, every node refined (subprog_binary.cpp:165-185)
    auto tree = mara::create_vertex_quadtree([] (std::size_t, double) { return true; }, bs, P.depth)
    .map([R] (auto block) { return (block * R).shared(); });
    P.xv.assign(N + 1, 0.0);
    P.yv.assign(N + 1, 0.0);
    for (int bi = 0; bi < nb; ++bi)
        for (int bj = 0; bj < nb; ++bj)
        {
            auto block = tree.at(mara::tree_index_t<2>{std::size_t(P.depth), {{std::size_t(bi), std::size_t(bj)}}});
            for (int i = 0; i <= bs; ++i)
                for (int j = 0; j <= bs; ++j)
                {
                    double x = block(i, j)[0].value, y = block(i, j)[1].value;
                    if (bj == 0 && j == 0) P.xv[bi * bs + i] = x;
                    if (bi == 0 && i == 0) P.yv[bj * bs + j] = y;
                }
        }
    for (int bi = 0; bi < nb; ++bi) // the grid is a tensor product: every block agrees with the 1-d arrays
        for (int bj = 0; bj < nb; ++bj)
        {
            auto block = tree.at(mara::tree_index_t<2>{std::size_t(P.depth), {{std::size_t(bi), std::size_t(bj)}}});
            for (int i = 0; i <= bs; ++i)
                for (int j = 0; j <= bs; ++j)
                    if (block(i, j)[0].value != P.xv[bi * bs + i] || block(i, j)[1].value != P.yv[bj * bs + j]) { std::fprintf(stderr, "vertices are not a tensor product\n"); return 3; }
        }
    P.h = 2.0 * R / bs / (1 << P.depth);

    // solver data (subprog_binary_solver_data.cpp:20-102)
    P.u_init.resize(std::size_t(N) * N);
    P.br.resize(P.u_init.size());
    double min_dx = P.xv[1] - P.xv[0], min_dy = P.yv[1] - P.yv[0], max_v = 1.0;
    for (int i = 0; i < N; ++i) { min_dx = std::min(min_dx, P.xv[i + 1] - P.xv[i]); min_dy = std::min(min_dy, P.yv[i + 1] - P.yv[i]); }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
        {
            double xc = (P.xv[i] + P.xv[i + 1]) * 0.5, yc = (P.yv[j] + P.yv[j + 1]) * 0.5;
            auto p = disk_profile(P, xc, yc);
            P.u_init[std::size_t(i) * N + j] = P.qform ? from_ref_q(p.to_conserved_angmom_per_area(make_loc(xc, yc))) : from_ref(p.to_conserved_per_area());
            max_v = std::max(max_v, p.velocity_magnitude().value);
            double rc = std::pow(xc * xc + yc * yc, 0.5);
            P.br[std::size_t(i) * N + j] = P.get("buffer_damping_rate") * (1.0 + std::tanh(3.0 * (rc - R)));
        }
    P.gst = P.get("source_term_softening") * std::min(min_dx, min_dy);
    P.recommended_dt = std::min(min_dx, min_dy) / max_v * P.get("cfl_number");

    solution_t S;
    S.u = P.u_init;
    S.E_acc = mara::make_full_orbital_elements_with_zeros();
    S.E_grav = mara::make_full_orbital_elements_with_zeros();
    mara::orbital_elements_t el;
    el.total_mass = 1.0; el.separation = P.get("separation"); el.mass_ratio = P.get("mass_ratio"); el.eccentricity = P.get("eccentricity");
    S.E = mara::make_full_orbital_elements(el);

    dump(prefix + ".xv.f64", P.xv.data(), P.xv.size() * 8);
    dump(prefix + ".yv.f64", P.yv.data(), P.yv.size() * 8);
    dump(prefix + ".u_init.f64", P.u_init.data(), P.u_init.size() * 24);
    dump(prefix + ".br.f64", P.br.data(), P.br.size() * 8);

    std::vector<double> scalars;
    bool safe = int(P.get("safe_mode"));
    auto step_dt = [&] (const solution_t& s) { return int(P.get("fixed_dt")) ? P.recommended_dt : P.get("cfl_number") * maximum_timestep(P, s); };

    {   // one stage from the initial state
        double dt = step_dt(S);
        double tot[NTOT];
        auto S1 = advance_u(P, S, dt, safe, tot);
        dump(prefix + ".u_stage.f64", S1.u.data(), S1.u.size() * 24);
        std::vector<double> st = {dt, P.recommended_dt, maximum_timestep(P, S)};
        for (int k = 0; k < 10; ++k) st.push_back(S1.acc[k]);
        push_elements(st, S1.E_acc); push_elements(st, S1.E_grav); push_elements(st, S1.E);
        for (int k = 0; k < NTOT; ++k) st.push_back(tot[k]);
        auto B = mara::compute_two_body_state(S.E, S.time);
        for (auto b : {B.body1, B.body2})
            for (double v : {b.mass, b.position_x, b.position_y, b.velocity_x, b.velocity_y}) st.push_back(v);
        dump(prefix + ".stage_scalars.f64", st.data(), st.size() * 8);
    }
    for (int n = 0; n < int(P.get("nsteps")); ++n)
    {
        double dt = step_dt(S);
        scalars.push_back(dt);
        if (int(P.get("rk_order")) == 1)
            S = advance_u(P, S, dt, safe);
        else
            S = combine(S, advance_u(P, advance_u(P, S, dt, safe), dt, safe));
    }
    dump(prefix + ".u_final.f64", S.u.data(), S.u.size() * 24);
    std::vector<double> fin = {S.time, double(S.iteration.as_integral())};
    for (int k = 0; k < 10; ++k) fin.push_back(S.acc[k]);
    push_elements(fin, S.E_acc); push_elements(fin, S.E_grav); push_elements(fin, S.E);
    for (double d : scalars) fin.push_back(d);
    dump(prefix + ".scalars.f64", fin.data(), fin.size() * 8);
    return 0;
}
