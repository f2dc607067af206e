/**
 * TEST INFRASTRUCTURE — reference-side oracle driver for `mara binary` on a GRADED block tree:
 * the default configuration of the sub-program (depth=4 block_size=24 focus_factor=2 focus_index=2) refines towards the
 * origin, so blocks of different levels meet: guard zones are prolonged / restricted and fluxes are corrected at
 * coarse-fine faces (src/subprog_binary_scheme.cpp:132-142, :614-720).
 *
 * As binary_ref.cpp (uniform trees), this driver composes reference HEADER functions, here including the tree machinery:
 *   mara::create_vertex_quadtree + ensure_valid_quadtree      mesh_tree_operators.hpp:115-190   (the real graded tree)
 *   mara::get_cell_block (refine_cells / combine_cells / coarsen_cells)   mesh_tree_operators.hpp:223-258, mesh_prolong_restrict.hpp
 *   mara::restrict_extrinsic                                   mesh_prolong_restrict.hpp:134-142
 *   arithmetic_binary_tree_t::{indexes, map, at, contains, node_at, sum}   core_tree.hpp
 *   iso2d recover_primitive / plm_gradient / riemann_hlle / angular_momentum / max_wavespeed, the two-body model
 * The arithmetic the reference writes inline in its translation unit is restated (same lines as cited in binary_ref.cpp); the
 * guard-zone gather and the flux correction follow scheme.cpp:132-142 and :614-700 call for call.
 *
 * usage: binary_tree_ref <out_prefix> [key=value ...]      (keys as binary_ref, plus focus_factor, focus_index)
 * writes <out_prefix>.{blocks.i32, xv, u_init, br, u_stage, u_final, stage_scalars, scalars, diag_scalars, diag_fields}; block data are concatenated in
 * tree order (arithmetic_binary_tree_t::sink), each block [bs][bs][3] (or [bs+1][2] interleaved x, y for the vertex edges).
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
using vec3   = mara::arithmetic_sequence_t<double, 3>;
using index_t = mara::tree_index_t<2>;
template<typename T> using tree_of = mara::arithmetic_binary_tree_t<nd::shared_array<T, 2>, 2>;

static cons_t to_ref(vec3 u)
{
    return cons_t().set<0>(mara::make_dimensional<-2, 1, 0>(u[0])).set<1>(mara::make_dimensional<-1, 1, -1>(u[1])).set<2>(mara::make_dimensional<-1, 1, -1>(u[2]));
}
static vec3 from_ref(const cons_t& U) { return {{mara::get<0>(U).value, mara::get<1>(U).value, mara::get<2>(U).value}}; }
static loc_t make_loc(double x, double y) { return {{mara::make_length(x), mara::make_length(y)}}; }
using consq_t = mara::iso2d::conserved_angmom_per_area_t;
static consq_t to_ref_q(vec3 q)
{
    return consq_t().set<0>(mara::make_dimensional<-2, 1, 0>(q[0])).set<1>(mara::make_dimensional<0, 1, -1>(q[1])).set<2>(mara::make_dimensional<0, 1, -1>(q[2]));
}
static vec3 from_ref_q(const consq_t& Q) { return {{mara::get<0>(Q).value, mara::get<1>(Q).value, mara::get<2>(Q).value}}; }

enum { T_MASS_ACC = 0, T_L_ACC = 2, T_TORQUE = 4, T_PX_ACC = 6, T_PY_ACC = 8, T_FX = 10, T_FY = 12, T_WORK = 14, T_MASS_EJ = 16, T_L_EJ = 17, NTOT = 18 };

struct params_t
{
    std::map<std::string, double> cfg;
    double get(const char* k) const { return cfg.at(k); }
    int bs = 0, depth = 0;
    mara::amr_types::vertex_2d_tree_t vertices;
    tree_of<vec3> u_init;
    tree_of<double> br;
    double recommended_dt = 0, gst = 0;
    bool qform = false;      // conserve_linear_p == 0: advance_q (scheme.cpp:906-1020), fields are (Sigma, Sigma s_r, Sigma l_z)
    double spacing(const index_t& i) const { return 2.0 * get("domain_radius") / bs / (1 << i.level); }
};

struct solution_t
{
    double time = 0;
    mara::rational_number_t iteration = 0;
    tree_of<vec3> u;
    double acc[10] = {0};
    mara::full_orbital_elements_t E_acc, E_grav, E;
};

template<typename T, typename F>
static nd::shared_array<T, 2> build(std::size_t n, std::size_t m, F f)
{
    auto a = nd::make_unique_array<T>(nd::make_shape(n, m));
    for (std::size_t i = 0; i < n; ++i)
        for (std::size_t j = 0; j < m; ++j)
            a(i, j) = f(i, j);
    return std::move(a).shared();
}

// guard zones: scheme.cpp:132-142
template<typename TreeType>
static auto extend(TreeType tree, std::size_t axis, std::size_t guard_count)
{
    return tree.indexes().map([tree, axis, guard_count] (auto index)
    {
        auto lower = mara::get_cell_block(tree, index.prev_on(axis), [=] (auto a) { return a | nd::select_final(guard_count, axis) | nd::to_shared(); });
        auto upper = mara::get_cell_block(tree, index.next_on(axis), [=] (auto a) { return a | nd::select_first(guard_count, axis) | nd::to_shared(); });
        return lower | nd::concat(tree.at(index)).on_axis(axis) | nd::concat(upper).on_axis(axis) | nd::to_shared();
    });
}

static prim_t disk_profile(const params_t& P, double x, double y)
{
    double rs = P.get("softening_radius"), rc = P.get("disk_radius"), Ma = P.get("mach_number");
    double s0 = P.get("disk_mass") / (17.0618 * rc * rc);
    double s1 = P.get("ambient_density") * s0;
    double mdot = P.get("mdot");
    auto sigma = [=] (double r) { auto q = r / rc; return s0 * std::exp(-0.5 * (q - 1) * (q - 1)) + s1; };
    auto dp_dr = [=] (double r) { auto q = r / rc; return (1.0 / Ma / Ma / (r + rs)) * (q * (1 - q) * (1 - s1 / sigma(r)) - 1.0); };
    double r = std::sqrt(x * x + y * y);
    double vp = std::sqrt(1.0 / (r + rs) + dp_dr(r)) * (int(P.get("counter_rotate")) ? -1 : 1);
    double vr = -mdot / (sigma(r) * 2 * M_PI * r) * (r > 2.0);
    return prim_t().with_sigma(sigma(r)).with_velocity_x(vr * (x / r) + vp * (-y / r)).with_velocity_y(vr * (y / r) + vp * (x / r));
}

static double phi_soft(const params_t& P, double x, double y, const mara::point_mass_t& b)
{
    double d0 = x - b.position_x, d1 = y - b.position_y;
    double rs2 = P.get("softening_radius") * P.get("softening_radius");
    return -1.0 * b.mass / std::pow(d0 * d0 + d1 * d1 + rs2, 0.5);
}
static double cs2_at(const params_t& P, double x, double y, const mara::two_body_state_t& B)
{
    double M = P.get("mach_number");
    if (int(P.get("axisymmetric_cs2"))) return 1.0 / std::sqrt(x * x + y * y) / M / M;
    return -(phi_soft(P, x, y, B.body1) + phi_soft(P, x, y, B.body2)) / M / M;
}
static double nu_at(const params_t& P, double x, double y, double cs2)
{
    double radius = std::sqrt(x * x + y * y);
    double rc = P.get("alpha_cutoff_radius");
    double profile = rc > 0.0 ? 0.5 * (1.0 + std::tanh(3.0 * (radius - rc))) : 1.0;
    if (P.get("nu") > 0.0) return profile * P.get("nu");
    return profile * P.get("alpha") * std::sqrt(cs2) * (radius / P.get("mach_number"));
}

static vec3 face_flux(const params_t& P, int axis, double h, double xf, double yf, const mara::two_body_state_t& B,
    prim_t pl, prim_t pr, prim_t gl, prim_t gr, prim_t hl, prim_t hr)
{
    auto pl_hat = pl + gl * 0.5 * h;
    auto pr_hat = pr - gr * 0.5 * h;
    double cs2 = cs2_at(P, xf, yf, B);
    double nu = nu_at(P, xf, yf, cs2);
    double mu = 0.5 * nu * (pl_hat.sigma() + pr_hat.sigma());
    auto F = mara::iso2d::riemann_hlle(pl_hat, pr_hat, cs2, cs2, mara::unit_vector_t::on_axis(std::size_t(axis)));
    double f0 = mara::get<0>(F).value, f1 = mara::get<1>(F).value, f2 = mara::get<2>(F).value;
    double v1, v2;
    if (axis == 0)
    {
        double dx_ux = 0.5 * (gl.velocity_x() + gr.velocity_x()), dx_uy = 0.5 * (gl.velocity_y() + gr.velocity_y());
        double dy_ux = 0.5 * (hl.velocity_x() + hr.velocity_x()), dy_uy = 0.5 * (hl.velocity_y() + hr.velocity_y());
        v1 = -(mu * (dx_ux - dy_uy));
        v2 = -(mu * (dx_uy + dy_ux));
    }
    else
    {
        double dx_ux = 0.5 * (hl.velocity_x() + hr.velocity_x()), dx_uy = 0.5 * (hl.velocity_y() + hr.velocity_y());
        double dy_ux = 0.5 * (gl.velocity_x() + gr.velocity_x()), dy_uy = 0.5 * (gl.velocity_y() + gr.velocity_y());
        v1 = -(mu * (dx_uy + dy_ux));
        v2 = -(-mu * (dx_ux - dy_uy));
    }
    vec3 f = {{f0 + 0.0, f1 + v1, f2 + v2}};
    if (P.qform)
    {
        double rd = P.get("domain_radius");
        double flux_sr = xf * f[1] + yf * f[2];
        double flux_lz = xf * f[2] - yf * f[1];
        if (axis == 0 && (xf == -rd || xf == rd)) flux_lz = 0.0;
        if (axis == 1 && (yf == -rd || yf == rd)) flux_lz = 0.0;
        f = {{f[0], flux_sr, flux_lz}};
    }
    return f;
}

// flux correction at refinement jumps, scheme.cpp:614-700: where the neighbour across a face is refined, the coarse face flux is
// replaced by the sum of the two fine fluxes (fluxes are already multiplied by their face length)
static tree_of<vec3> correct_fluxes(const tree_of<vec3>& fhat, std::size_t axis)
{
    const std::size_t other = 1 - axis;
    return fhat.indexes().map([&fhat, axis, other] (auto index)
    {
        auto f = fhat.at(index);
        auto lo = index.prev_on(axis), hi = index.next_on(axis);
        if (! fhat.contains(lo) && ! fhat.contains(lo.parent_index()))
        {
            auto node = fhat.node_at(lo);
            index_t c0 = {1, {{0, 0}}}, c1 = {1, {{0, 0}}};
            c0.coordinates[axis] = 1; c0.coordinates[other] = 0;
            c1.coordinates[axis] = 1; c1.coordinates[other] = 1;
            auto fine = node.at(c0) | nd::concat(node.at(c1)).on_axis(other) | mara::restrict_extrinsic(other) | nd::take_final_on_axis(axis);
            f = fine | nd::concat(f | nd::drop_first_on_axis(axis)).on_axis(axis) | nd::to_shared();
        }
        if (! fhat.contains(hi) && ! fhat.contains(hi.parent_index()))
        {
            auto node = fhat.node_at(hi);
            index_t c0 = {1, {{0, 0}}}, c1 = {1, {{0, 0}}};
            c0.coordinates[axis] = 0; c0.coordinates[other] = 0;
            c1.coordinates[axis] = 0; c1.coordinates[other] = 1;
            auto fine = node.at(c0) | nd::concat(node.at(c1)).on_axis(other) | mara::restrict_extrinsic(other) | nd::take_first_on_axis(axis);
            f = (f | nd::drop_final_on_axis(axis)) | nd::concat(fine).on_axis(axis) | nd::to_shared();
        }
        return f;
    });
}

static solution_t advance_u(const params_t& P, const solution_t& S, double dt, bool safe_mode, double* totals_out = nullptr)
{
    const std::size_t bs = P.bs;
    const double th = safe_mode ? 0.0 : P.get("plm_theta");
    auto B = mara::compute_two_body_state(S.E, S.time);
    const mara::point_mass_t bodies[2] = {B.body1, B.body2};

    auto cell_center = [] (auto xv, std::size_t i, std::size_t j)
    {
        double xc = (((xv(i, j)[0] + xv(i + 1, j)[0]) * 0.5 + (xv(i, j + 1)[0] + xv(i + 1, j + 1)[0]) * 0.5) * 0.5).value;
        double yc = (((xv(i, j)[1] + xv(i + 1, j)[1]) * 0.5 + (xv(i, j + 1)[1] + xv(i + 1, j + 1)[1]) * 0.5) * 0.5).value;
        return make_loc(xc, yc);
    };
    auto p0 = S.u.pair(P.vertices).map([&] (auto ux) { auto U = ux.first; auto xv = ux.second;
        return build<prim_t>(bs, bs, [&] (auto i, auto j) { return P.qform ? mara::iso2d::recover_primitive(to_ref_q(U(i, j)), cell_center(xv, i, j))
                                                                            : mara::iso2d::recover_primitive(to_ref(U(i, j))); }); });
    auto p0_ex = extend(p0, 0, 1);
    auto p0_ey = extend(p0, 1, 1);
    auto gx = p0_ex.pair(p0_ex.indexes()).map([&] (auto pi) { auto p = pi.first; double h = P.spacing(pi.second);
        return build<prim_t>(bs, bs, [&] (auto i, auto j) { return mara::plm_gradient(p(i, j), p(i + 1, j), p(i + 2, j), th) / h; }); });
    auto gy = p0_ey.pair(p0_ey.indexes()).map([&] (auto pi) { auto p = pi.first; double h = P.spacing(pi.second);
        return build<prim_t>(bs, bs, [&] (auto i, auto j) { return mara::plm_gradient(p(i, j), p(i, j + 1), p(i, j + 2), th) / h; }); });
    auto gx_ex = extend(gx, 0, 1), gx_ey = extend(gx, 1, 1), gy_ex = extend(gy, 0, 1), gy_ey = extend(gy, 1, 1);

    // block_fluxes_u :472-516
    auto fhat_x = p0.indexes().map([&] (auto index)
    {
        auto xv = P.vertices.at(index); auto pe = p0_ex.at(index); auto gl = gx_ex.at(index); auto gt = gy_ex.at(index);
        double h = P.spacing(index);
        return build<vec3>(bs + 1, bs, [&] (auto i, auto j)
        {
            double xf = ((xv(i, j)[0] + xv(i, j + 1)[0]) * 0.5).value, yf = ((xv(i, j)[1] + xv(i, j + 1)[1]) * 0.5).value;
            auto f = face_flux(P, 0, h, xf, yf, B, pe(i, j), pe(i + 1, j), gl(i, j), gl(i + 1, j), gt(i, j), gt(i + 1, j));
            double dy = (xv(i, j + 1)[1] - xv(i, j)[1]).value;
            return vec3{{f[0] * dy, f[1] * dy, f[2] * dy}};
        });
    });
    auto fhat_y = p0.indexes().map([&] (auto index)
    {
        auto xv = P.vertices.at(index); auto pe = p0_ey.at(index); auto gl = gy_ey.at(index); auto gt = gx_ey.at(index);
        double h = P.spacing(index);
        return build<vec3>(bs, bs + 1, [&] (auto i, auto j)
        {
            double xf = ((xv(i, j)[0] + xv(i + 1, j)[0]) * 0.5).value, yf = ((xv(i, j)[1] + xv(i + 1, j)[1]) * 0.5).value;
            auto f = face_flux(P, 1, h, xf, yf, B, pe(i, j), pe(i, j + 1), gl(i, j), gl(i, j + 1), gt(i, j), gt(i, j + 1));
            double dx = (xv(i + 1, j)[0] - xv(i, j)[0]).value;
            return vec3{{f[0] * dx, f[1] * dx, f[2] * dx}};
        });
    });
    auto fx = correct_fluxes(fhat_x, 0);
    auto fy = correct_fluxes(fhat_y, 1);

    // block_update_u :568-587 with source_terms_u :345-411
    const double rs2 = P.get("softening_radius") * P.get("softening_radius"), s2 = P.get("sink_radius") * P.get("sink_radius");
    const double floor_sigma = P.get("density_floor") * P.get("disk_mass");
    bool negative = false;
    std::map<std::pair<std::size_t, std::pair<std::size_t, std::size_t>>, std::vector<double>> block_totals;
    auto key_of = [] (const index_t& i) { return std::make_pair(i.level, std::make_pair(i.coordinates[0], i.coordinates[1])); };

    auto u1 = p0.indexes().map([&] (auto index)
    {
        auto xv = P.vertices.at(index); auto U0 = S.u.at(index); auto Ui = P.u_init.at(index); auto Br = P.br.at(index);
        auto FX = fx.at(index); auto FY = fy.at(index); auto Pc = p0.at(index);
        std::vector<double> t(NTOT, 0.0);
        double sink_sum[2][3] = {{0}};
        auto out = build<vec3>(bs, bs, [&] (auto i, auto j)
        {
            // cell centres and areas as create_solver_data :22-36
            double xc = (((xv(i, j)[0] + xv(i + 1, j)[0]) * 0.5 + (xv(i, j + 1)[0] + xv(i + 1, j + 1)[0]) * 0.5) * 0.5).value;
            double yc = (((xv(i, j)[1] + xv(i + 1, j)[1]) * 0.5 + (xv(i, j + 1)[1] + xv(i + 1, j + 1)[1]) * 0.5) * 0.5).value;
            double dxm = (((xv(i + 1, j)[0] - xv(i, j)[0]) + (xv(i + 1, j + 1)[0] - xv(i, j + 1)[0])) * 0.5).value;
            double dym = (((xv(i, j + 1)[1] - xv(i, j)[1]) + (xv(i + 1, j + 1)[1] - xv(i + 1, j)[1])) * 0.5).value;
            double dA = dxm * dym;
            vec3 u0 = U0(i, j);
            auto lz = [xc, yc] (vec3 u) { return mara::iso2d::angular_momentum(to_ref(u), make_loc(xc, yc)).value; };
            vec3 s_grav[2], s_sink[2];
            double fg[2][2];
            for (int b = 0; b < 2; ++b)
            {
                double d0 = xc - bodies[b].position_x, d1 = yc - bodies[b].position_y;
                double den = std::pow(d0 * d0 + d1 * d1 + rs2, 1.5);
                fg[b][0] = (-d0 / den * 1.0 * bodies[b].mass) * u0[0];
                fg[b][1] = (-d1 / den * 1.0 * bodies[b].mass) * u0[0];
                s_grav[b] = {{0.0 * dt, fg[b][0] * dt, fg[b][1] * dt}};
                double a2 = (d0 * d0 + d1 * d1) / s2 / 2.0;
                double rate = P.get("sink_rate") * std::exp(-a2);
                s_sink[b] = {{-u0[0] * rate * dt, -u0[1] * rate * dt, -u0[2] * rate * dt}};
            }
            vec3 ui = Ui(i, j);
            double br = Br(i, j);
            vec3 s_buffer = {{(ui[0] - u0[0]) * br * dt, (ui[1] - u0[1]) * br * dt, (ui[2] - u0[2]) * br * dt}};
            double fl = double(u0[0] < floor_sigma);
            vec3 s_floor = {{u0[0] * 1e-2 * fl, u0[1] * 1e-2 * fl, u0[2] * 1e-2 * fl}};
            vec3 dps[2] = {s_sink[0], s_sink[1]};
            if (P.qform)
            {
                // source_terms_q :417-466
                for (int b = 0; b < 2; ++b)
                {
                    s_grav[b] = {{0.0 * dt, (xc * fg[b][0] + yc * fg[b][1]) * dt, (xc * fg[b][1] - yc * fg[b][0]) * dt}};
                    dps[b] = from_ref(mara::iso2d::to_conserved_per_area(to_ref_q(s_sink[b]), make_loc(xc, yc)));
                }
                double sr2 = std::pow(P.gst, 2.0);
                double ramp = 1.0 - std::exp(-(xc * xc + yc * yc) / sr2);
                auto sg = Pc(i, j).source_terms_conserved_angmom(cs2_at(P, xc, yc, B));
                s_floor = {{mara::get<0>(sg).value * ramp * dt, mara::get<1>(sg).value * ramp * dt, mara::get<2>(sg).value * ramp * dt}};
            }
            for (int b = 0; b < 2; ++b)
            {
                t[T_MASS_ACC + b] = t[T_MASS_ACC + b] + s_sink[b][0] * dA;
                t[T_L_ACC + b]    = t[T_L_ACC + b] + (P.qform ? s_sink[b][2] : lz(s_sink[b])) * dA;
                t[T_TORQUE + b]   = t[T_TORQUE + b] + (P.qform ? s_grav[b][2] : lz(s_grav[b])) * dA;
                t[T_FX + b]       = t[T_FX + b] + fg[b][0] * dt * dA;
                t[T_FY + b]       = t[T_FY + b] + fg[b][1] * dt * dA;
                t[T_PX_ACC + b]   = t[T_PX_ACC + b] + dps[b][1] * dA;
                t[T_PY_ACC + b]   = t[T_PY_ACC + b] + dps[b][2] * dA;
                for (int q = 0; q < 3; ++q) sink_sum[b][q] = sink_sum[b][q] + s_sink[b][q] * dA;
            }
            t[T_L_EJ]    = t[T_L_EJ] + (P.qform ? s_buffer[2] : lz(s_buffer)) * dA;
            t[T_MASS_EJ] = t[T_MASS_EJ] + s_buffer[0] * dA;
            vec3 r;
            for (int q = 0; q < 3; ++q)
            {
                double l = (FX(i + 1, j)[q] - FX(i, j)[q]) + (FY(i, j + 1)[q] - FY(i, j)[q]);
                double s = s_grav[0][q] + s_grav[1][q] + s_sink[0][q] + s_sink[1][q] + s_buffer[q] + s_floor[q];
                r[q] = u0[q] - l * dt / dA + s;
            }
            if (r[0] < 0.0) negative = true;
            return r;
        });
        for (int k = 0; k < NTOT; ++k) t[k] = -t[k];
        for (int b = 0; b < 2; ++b)
        {
            double M0 = bodies[b].mass, px0 = bodies[b].velocity_x * M0, py0 = bodies[b].velocity_y * M0;
            double M1 = M0 + -sink_sum[b][0], px1 = px0 + -sink_sum[b][1], py1 = py0 + -sink_sum[b][2];
            t[T_WORK + b] = ((px1 * px1 + py1 * py1) / M1 - (px0 * px0 + py0 * py0) / M0) * 0.5;
            if (P.qform) t[T_WORK + b] = 0.0;
        }
        block_totals[key_of(index)] = t;
        return out;
    });
    if (negative) throw std::runtime_error("negative density in updated state");

    double tot[NTOT];
    for (int k = 0; k < NTOT; ++k)
        tot[k] = p0.indexes().map([&] (auto index) { return block_totals.at(key_of(index))[k]; }).sum();     // the tree's own fold order

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

    solution_t R = S;
    R.u = u1;
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
    solution_t r;
    r.time = a.time * 0.5 + b.time * 0.5;
    r.iteration = a.iteration * mara::make_rational(1, 2) + b.iteration * mara::make_rational(1, 2);
    r.u = a.u.pair(b.u).map([] (auto ab) { auto A = ab.first; auto Bq = ab.second;
        return build<vec3>(A.shape(0), A.shape(1), [&] (auto i, auto j) { return vec3{{A(i, j)[0] * 0.5 + Bq(i, j)[0] * 0.5, A(i, j)[1] * 0.5 + Bq(i, j)[1] * 0.5, A(i, j)[2] * 0.5 + Bq(i, j)[2] * 0.5}}; }); });
    for (int k = 0; k < 10; ++k) r.acc[k] = a.acc[k] * 0.5 + b.acc[k] * 0.5;
    r.E_acc = a.E_acc * 0.5 + b.E_acc * 0.5;
    r.E_grav = a.E_grav * 0.5 + b.E_grav * 0.5;
    r.E = a.E * 0.5 + b.E * 0.5;
    return r;
}

static double maximum_timestep(const params_t& P, const solution_t& S)
{
    auto B = mara::compute_two_body_state(S.E, S.time);
    const std::size_t bs = P.bs;
    return S.u.pair(S.u.indexes()).map([&] (auto ui)
    {
        auto U = ui.first; auto index = ui.second; auto xv = P.vertices.at(index);
        double a = 0.0; bool first = true;
        for (std::size_t i = 0; i < bs; ++i)
            for (std::size_t j = 0; j < bs; ++j)
            {
                double xc = (((xv(i, j)[0] + xv(i + 1, j)[0]) * 0.5 + (xv(i, j + 1)[0] + xv(i + 1, j + 1)[0]) * 0.5) * 0.5).value;
                double yc = (((xv(i, j)[1] + xv(i + 1, j)[1]) * 0.5 + (xv(i, j + 1)[1] + xv(i + 1, j + 1)[1]) * 0.5) * 0.5).value;
                auto pc = P.qform ? mara::iso2d::recover_primitive(to_ref_q(U(i, j)), make_loc(xc, yc)) : mara::iso2d::recover_primitive(to_ref(U(i, j)));
                double w = pc.max_wavespeed(cs2_at(P, xc, yc, B));
                a = first ? w : std::max(a, w); first = false;
            }
        return P.spacing(index) / a;
    }).min();
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
template<typename T> static std::vector<double> flatten(const tree_of<T>& tree)
{
    std::vector<double> out;
    tree.sink([&out] (auto block) { for (auto v : block) { if constexpr (std::is_same<T, double>::value) out.push_back(v); else for (int q = 0; q < 3; ++q) out.push_back(v[q]); } });
    return out;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 1;
    std::string prefix = argv[1];
    params_t P;
    P.cfg = {
        {"cfl_number", 0.4}, {"fixed_dt", 0}, {"depth", 4}, {"begin_live_binary", 1e6}, {"block_size", 24}, {"rk_order", 2},
        {"plm_theta", 1.8}, {"source_term_softening", 1.}, {"softening_radius", 0.05}, {"sink_radius", 0.05}, {"sink_rate", 1.0},
        {"buffer_damping_rate", 10.0}, {"domain_radius", 12.0}, {"disk_radius", 2.0}, {"disk_mass", 1e-3}, {"ambient_density", 1e-4},
        {"density_floor", 0.0}, {"separation", 1.0}, {"mass_ratio", 1.0}, {"eccentricity", 0.0}, {"counter_rotate", 0},
        {"mach_number", 10.0}, {"axisymmetric_cs2", 0}, {"no_accretion_force", 0}, {"alpha_cutoff_radius", 0.0}, {"alpha", 0.1},
        {"nu", 0.0}, {"mdot", 0.0}, {"nsteps", 1}, {"safe_mode", 0}, {"focus_factor", 2.0}, {"focus_index", 2.0}, {"conserve_linear_p", 1}};
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
    const std::size_t bs = P.bs;
    const double R = P.get("domain_radius");
    const double ff = P.get("focus_factor"), fi = P.get("focus_index");

    // subprog_binary.cpp:165-185
    P.vertices = mara::create_vertex_quadtree([ff, fi] (std::size_t level, double centroid_radius) { return centroid_radius < ff / std::pow(level, fi); }, bs, P.depth)
    .map([R] (auto block) { return (block * R).shared(); });

    // solver data, subprog_binary_solver_data.cpp:20-102
    auto centers = [] (auto xv, std::size_t i, std::size_t j)
    {
        double xc = (((xv(i, j)[0] + xv(i + 1, j)[0]) * 0.5 + (xv(i, j + 1)[0] + xv(i + 1, j + 1)[0]) * 0.5) * 0.5).value;
        double yc = (((xv(i, j)[1] + xv(i + 1, j)[1]) * 0.5 + (xv(i, j + 1)[1] + xv(i + 1, j + 1)[1]) * 0.5) * 0.5).value;
        return std::make_pair(xc, yc);
    };
    P.u_init = P.vertices.map([&] (auto xv) { return build<vec3>(bs, bs, [&] (auto i, auto j) { auto c = centers(xv, i, j); auto p = disk_profile(P, c.first, c.second);
        return P.qform ? from_ref_q(p.to_conserved_angmom_per_area(make_loc(c.first, c.second))) : from_ref(p.to_conserved_per_area()); }); });
    P.br = P.vertices.map([&] (auto xv) { return build<double>(bs, bs, [&] (auto i, auto j) { auto c = centers(xv, i, j);
        double rc = std::pow(c.first * c.first + c.second * c.second, 0.5); return P.get("buffer_damping_rate") * (1.0 + std::tanh(3.0 * (rc - R))); }); });
    double min_dx = 1e300, min_dy = 1e300, max_v = 1.0;
    P.vertices.sink([&] (auto xv)
    {
        for (std::size_t i = 0; i < bs; ++i)
            for (std::size_t j = 0; j <= bs; ++j)
            {
                min_dx = std::min(min_dx, (xv(i + 1, j)[0] - xv(i, j)[0]).value);
                min_dy = std::min(min_dy, (xv(j, i + 1)[1] - xv(j, i)[1]).value);
            }
        for (std::size_t i = 0; i < bs; ++i)
            for (std::size_t j = 0; j < bs; ++j)
            {
                auto c = centers(xv, i, j);
                max_v = std::max(max_v, disk_profile(P, c.first, c.second).velocity_magnitude().value);
            }
    });
    P.recommended_dt = std::min(min_dx, min_dy) / max_v * P.get("cfl_number");
    P.gst = P.get("source_term_softening") * std::min(min_dx, min_dy);

    solution_t S;
    S.u = P.u_init;
    S.E_acc = mara::make_full_orbital_elements_with_zeros();
    S.E_grav = mara::make_full_orbital_elements_with_zeros();
    mara::orbital_elements_t el;
    el.total_mass = 1.0; el.separation = P.get("separation"); el.mass_ratio = P.get("mass_ratio"); el.eccentricity = P.get("eccentricity");
    S.E = mara::make_full_orbital_elements(el);

    std::vector<int> blocks;
    P.vertices.indexes().sink([&blocks] (auto index) { blocks.push_back(int(index.level)); blocks.push_back(int(index.coordinates[0])); blocks.push_back(int(index.coordinates[1])); });
    dump(prefix + ".blocks.i32", blocks.data(), blocks.size() * 4);
    std::vector<double> edges;      // per block: x of vertices (i, 0), i = 0..bs, then y of vertices (0, j)
    P.vertices.sink([&] (auto xv) { for (std::size_t i = 0; i <= bs; ++i) edges.push_back(xv(i, 0)[0].value); for (std::size_t j = 0; j <= bs; ++j) edges.push_back(xv(0, j)[1].value); });
    dump(prefix + ".xv.f64", edges.data(), edges.size() * 8);
    auto ui = flatten(P.u_init); dump(prefix + ".u_init.f64", ui.data(), ui.size() * 8);
    auto br = flatten(P.br); dump(prefix + ".br.f64", br.data(), br.size() * 8);

    bool safe = int(P.get("safe_mode"));
    auto step_dt = [&] (const solution_t& s) { return int(P.get("fixed_dt")) ? P.recommended_dt : P.get("cfl_number") * maximum_timestep(P, s); };
    {
        double dt = step_dt(S);
        double tot[NTOT];
        auto S1 = advance_u(P, S, dt, safe, tot);
        auto us = flatten(S1.u); dump(prefix + ".u_stage.f64", us.data(), us.size() * 8);
        std::vector<double> st = {dt, P.recommended_dt, maximum_timestep(P, S)};
        for (int k = 0; k < 10; ++k) st.push_back(S1.acc[k]);
        push_elements(st, S1.E_acc); push_elements(st, S1.E_grav); push_elements(st, S1.E);
        for (int k = 0; k < NTOT; ++k) st.push_back(tot[k]);
        auto B = mara::compute_two_body_state(S.E, S.time);
        for (auto b : {B.body1, B.body2})
            for (double v : {b.mass, b.position_x, b.position_y, b.velocity_x, b.velocity_y}) st.push_back(v);
        dump(prefix + ".stage_scalars.f64", st.data(), st.size() * 8);
    }
    std::vector<double> dts;
    for (int n = 0; n < int(P.get("nsteps")); ++n)
    {
        double dt = step_dt(S);
        dts.push_back(dt);
        if (int(P.get("rk_order")) == 1) S = advance_u(P, S, dt, safe);
        else                            S = combine(S, advance_u(P, advance_u(P, S, dt, safe), dt, safe));
    }
    auto uf = flatten(S.u); dump(prefix + ".u_final.f64", uf.data(), uf.size() * 8);
    std::vector<double> fin = {S.time, double(S.iteration.as_integral())};
    for (int k = 0; k < 10; ++k) fin.push_back(S.acc[k]);
    push_elements(fin, S.E_acc); push_elements(fin, S.E_grav); push_elements(fin, S.E);
    for (double d : dts) fin.push_back(d);
    dump(prefix + ".scalars.f64", fin.data(), fin.size() * 8);

    // Diagnostics of the final state, composed as subprog_binary_diagnostics.cpp:21-82 composes them:
    // disk_mass, disk_angular_momentum (time series) and sigma, radial_velocity, phi_velocity (diagnostics file).
    {
        auto dA = P.vertices.map([] (auto block)                // subprog_binary_solver_data.cpp:29-34
        {
            auto dx = block | nd::map([] (auto p) { return p[0]; }) | nd::difference_on_axis(0) | nd::midpoint_on_axis(1);
            auto dy = block | nd::map([] (auto p) { return p[1]; }) | nd::difference_on_axis(1) | nd::midpoint_on_axis(0);
            return (dx * dy) | nd::to_shared();
        });
        auto cc = P.vertices.map([] (auto block) { return block | nd::midpoint_on_axis(0) | nd::midpoint_on_axis(1) | nd::to_shared(); });
        double disk_mass = 0, disk_lz = 0;
        std::vector<double> fields;                              // per block: sigma [bs][bs], then v_r, then v_phi
        auto push_fields = [&] (auto p0)
        {
            auto xc = cc.map([] (auto b) { return b | nd::map([] (auto x) { return x[0]; }) | nd::to_shared(); });
            auto yc = cc.map([] (auto b) { return b | nd::map([] (auto x) { return x[1]; }) | nd::to_shared(); });
            auto rc = (xc * xc + yc * yc).map(nd::map([] (mara::unit_area<double> r2) { return r2.pow<1, 2>(); }));
            auto rhat_x =  xc / rc;
            auto rhat_y =  yc / rc;
            auto phat_x = -yc / rc;
            auto phat_y =  xc / rc;
            auto sigma = p0.map(nd::map(std::mem_fn(&prim_t::sigma)));
            auto vx    = p0.map(nd::map(std::mem_fn(&prim_t::velocity_x)));
            auto vy    = p0.map(nd::map(std::mem_fn(&prim_t::velocity_y)));
            auto vr    = vx * rhat_x + vy * rhat_y;
            auto vp    = vx * phat_x + vy * phat_y;
            auto s_ = sigma.map(nd::to_shared()), r_ = vr.map(nd::to_shared()), p_ = vp.map(nd::to_shared());
            s_.indexes().sink([&] (auto index)
            {
                for (auto v : s_.at(index)) fields.push_back(v);
                for (auto v : r_.at(index)) fields.push_back(v);
                for (auto v : p_.at(index)) fields.push_back(v);
            });
        };
        if (! P.qform)
        {
            auto U = S.u.map([] (auto b) { return b | nd::map([] (vec3 u) { return to_ref(u); }) | nd::to_shared(); });
            disk_mass = (U.map(nd::map([] (auto u) { return mara::get<0>(u); })) * dA).map(nd::sum()).sum().value;
            auto lz = U.pair(cc).apply([] (auto Ub, auto X) { return nd::zip(Ub, X) | nd::apply(mara::iso2d::angular_momentum); });
            disk_lz = (lz * dA).map(nd::sum()).sum().value;
            push_fields(U.map([] (auto Ub) { return Ub | nd::map([] (auto u) { return mara::iso2d::recover_primitive(u); }) | nd::to_shared(); }));
        }
        else
        {
            auto Q = S.u.map([] (auto b) { return b | nd::map([] (vec3 u) { return to_ref_q(u); }) | nd::to_shared(); });
            disk_mass = (Q.map(nd::map([] (auto q) { return mara::get<0>(q); })) * dA).map(nd::sum()).sum().value;
            disk_lz = (Q * dA).map(nd::map([] (auto q) { return mara::get<2>(q); })).map(nd::sum()).sum().value;
            push_fields(Q.pair(cc).apply([] (auto Qb, auto X) { return nd::zip(Qb, X) | nd::apply([] (auto q, auto x) { return mara::iso2d::recover_primitive(q, x); }) | nd::to_shared(); }));
        }
        auto B = mara::compute_two_body_state(S.E, S.time);
        std::vector<double> dg = {disk_mass, disk_lz, B.body1.position_x, B.body1.position_y, B.body2.position_x, B.body2.position_y};
        dump(prefix + ".diag_scalars.f64", dg.data(), dg.size() * 8);
        dump(prefix + ".diag_fields.f64", fields.data(), fields.size() * 8);
    }
    return 0;
}
