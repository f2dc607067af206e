// Host-side scalar bookkeeping of the `binary` sub-program's solution_t: what binary::advance_u does with the
// source-term totals of one stage (src/subprog_binary_scheme.cpp:832-902) and the scalar part of the Runge-Kutta
// combine s0 * 1/2 + s2 * 1/2 (:1033-1069). A few dozen flops per stage, kept on the host exactly as upstream;
// compiled with g++ (like twobody.cpp) so that libm calls inside the orbital-element fit are scheduled as in the
// reference build.
#include <cmath>
#include <vector>
#include <map>
#include <set>
#include <array>
#include <algorithm>
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

namespace mh {

// one tensor-product grid (the uniform mesh, or one block of a graded tree): initial field, buffer rate, smallest spacings, largest speed
void binary_grid_data(const mh_binary_model* m, int n, const double* xv, const double* yv, double* u_init, double* br, double* min_dx_out, double* min_dy_out, double* max_v_out)
{
    const double rs = m->softening_radius, rc = m->disk_radius, Ma = m->mach_number;
    const double s0 = m->disk_mass / (17.0618 * rc * rc);
    const double s1 = m->ambient_density * s0;
    auto sigma = [=] (double r) { const double x = r / rc; return s0 * std::exp(-0.5 * (x - 1) * (x - 1)) + s1; };
    auto dp_dr = [=] (double r) { const double x = r / rc; return (1.0 / Ma / Ma / (r + rs)) * (x * (1 - x) * (1 - s1 / sigma(r)) - 1.0); };
    double min_dx = xv[1] - xv[0], min_dy = yv[1] - yv[0], max_v = 0.0;
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
    *min_dx_out = min_dx;
    *min_dy_out = min_dy;
    *max_v_out = max_v;
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
    double min_dx, min_dy, max_v;
    mh::binary_grid_data(m, n, xv, yv, u_init, br, &min_dx, &min_dy, &max_v);
    *recommended_dt = std::fmin(min_dx, min_dy) / std::fmax(1.0, max_v) * m->cfl_number;
    return MH_OK;
}

} // extern "C"

// ---- graded block tree ------------------------------------------------------------------------------------------------------
namespace {

using Key = std::array<int, 3>;      // level, i, j

struct LeafSet
{
    std::set<Key> leaves;
    static Key wrap(Key k) { const int n = 1 << k[0]; k[1] = (k[1] % n + n) % n; k[2] = (k[2] % n + n) % n; return k; }
    // the level difference to the deepest leaf below node k, or -1 if no node exists there (k lies inside a coarser leaf)
    int node_depth(const Key& k) const
    {
        int best = -1;
        for (const auto& l : leaves)
        {
            if (l[0] < k[0]) continue;
            const int s = l[0] - k[0];
            if ((l[1] >> s) == k[1] && (l[2] >> s) == k[2]) best = std::max(best, s);
        }
        return best;
    }
    void bifurcate(const Key& k)
    {
        leaves.erase(k);
        for (int c = 0; c < 4; ++c) leaves.insert({k[0] + 1, 2 * k[1] + (c & 1), 2 * k[2] + ((c >> 1) & 1)});
    }
    // leaves in the order arithmetic_binary_tree_t visits them: depth first, children in orthant order (core_tree.hpp:156-159)
    void ordered(const Key& node, std::vector<Key>& out) const
    {
        if (leaves.count(node)) { out.push_back(node); return; }
        if (node_depth(node) < 0) return;
        for (int c = 0; c < 4; ++c) ordered({node[0] + 1, 2 * node[1] + (c & 1), 2 * node[2] + ((c >> 1) & 1)}, out);
    }
};

// the level-L refinement of linspace(-1, 1, bs + 1): prolong_verts applied L times (mesh_prolong_restrict.hpp:148-159)
std::vector<double> level_vertices(int bs, int level)
{
    int n = bs;
    std::vector<double> a(n + 1), b;
    for (int i = 0; i <= n; ++i) a[i] = -1.0 + (1.0 - -1.0) * i / double((n + 1) - 1);
    for (int l = 0; l < level; ++l)
    {
        b.resize(2 * n + 1);
        for (int i = 0; i <= 2 * n; ++i) b[i] = (a[i / 2] + a[(i + 1) / 2]) * 0.5;
        n *= 2;
        a.swap(b);
    }
    return a;
}

} // namespace

namespace mh {

// neighbour table of the leaf blocks: topo[b][side][3] = {kind, id0, id1 | half}; sides: 0 = -x, 1 = +x, 2 = -y, 3 = +y.
// kind 0: a leaf of the same level (id0); 1: a coarser leaf (id0) of which this block touches half `half` of the edge;
// 2: two finer leaves id0, id1 in tangential order. Indices wrap periodically at every level (core_tree.hpp:203-204).
int binary_tree_topology(const mh_tree_block* blocks, int nb, int32_t* topo)
{
    std::map<Key, int> id;
    for (int b = 0; b < nb; ++b) id[{blocks[b].level, blocks[b].i, blocks[b].j}] = b;
    for (int b = 0; b < nb; ++b)
        for (int s = 0; s < 4; ++s)
        {
            const int axis = s / 2, up = s % 2;
            Key n = {blocks[b].level, blocks[b].i, blocks[b].j};
            const int tangential = axis == 0 ? n[2] : n[1];
            n[1 + axis] += up ? 1 : -1;
            n = LeafSet::wrap(n);
            int32_t* t = topo + ((std::size_t) b * 4 + s) * 3;
            auto it = id.find(n);
            if (it != id.end()) { t[0] = 0; t[1] = it->second; t[2] = 0; continue; }
            it = n[0] > 0 ? id.find({n[0] - 1, n[1] / 2, n[2] / 2}) : id.end();
            if (it != id.end()) { t[0] = 1; t[1] = it->second; t[2] = tangential & 1; continue; }
            // the two children of n that touch our face: on the far side of n along `axis` when n is below us, on the near side otherwise
            Key c0 = {n[0] + 1, 2 * n[1], 2 * n[2]}, c1 = c0;
            c0[1 + axis] += up ? 0 : 1;
            c1[1 + axis] += up ? 0 : 1;
            c1[2 - axis] += 1;
            auto i0 = id.find(c0), i1 = id.find(c1);
            if (i0 == id.end() || i1 == id.end()) { set_error("binary tree: block %d has a neighbour more than one level finer or coarser", b); return MH_E_INVALID; }
            t[0] = 2; t[1] = i0->second; t[2] = i1->second;
        }
    return MH_OK;
}

// Position of each leaf along the Hilbert curve of the finest level present: a leaf of level l at (i, j) covers an aligned square of
// 4^(D - l) finest cells, which the curve visits in one piece, so the curve index of ANY of its cells orders the leaves; the lower-left
// one is taken. order[k] = the block that stands k-th along the curve. (The reference declares a hilbert_index for tree indexes,
// src/core_tree.hpp:1033-1069, which no sub-program calls; it walks the bits below `level`, not below 2^level, so it cannot order a
// tree. This is the curve it names - https://en.wikipedia.org/wiki/Hilbert_curve, xy2d - applied to the leaves.)
int binary_tree_curve_order(const mh_tree_block* blocks, int nb, int32_t* order)
{
    int depth = 0;
    for (int b = 0; b < nb; ++b)
    {
        if (blocks[b].level < 0 || blocks[b].level > 30) { set_error("binary tree: level %d of block %d out of range", blocks[b].level, b); return MH_E_INVALID; }
        if (blocks[b].level > depth) depth = blocks[b].level;
    }
    // key = the first curve index of the leaf's square (the curve visits an aligned square of 4^(D - l) cells in one aligned run)
    std::vector<std::pair<uint64_t, int>> key(nb);
    for (int b = 0; b < nb; ++b)
    {
        const int shift = depth - blocks[b].level;
        uint64_t x = (uint64_t) blocks[b].i << shift, y = (uint64_t) blocks[b].j << shift, d = 0;
        const uint64_t n = (uint64_t) 1 << depth;
        if (blocks[b].i < 0 || blocks[b].j < 0 || x >= n || y >= n) { set_error("binary tree: block %d (%d, %d, %d) lies outside the domain", b, blocks[b].level, blocks[b].i, blocks[b].j); return MH_E_INVALID; }
        for (uint64_t s2 = n / 2; s2 > 0; s2 /= 2)
        {
            const uint64_t rx = (x & s2) > 0, ry = (y & s2) > 0;
            d += s2 * s2 * ((3 * rx) ^ ry);
            if (ry == 0)
            {
                if (rx == 1) { x = n - 1 - x; y = n - 1 - y; }
                const uint64_t tmp = x; x = y; y = tmp;
            }
        }
        const uint64_t cells = (uint64_t) 1 << (2 * shift);
        key[b] = {d - d % cells, b};
    }
    std::sort(key.begin(), key.end());
    for (int k = 0; k + 1 < nb; ++k)
    {
        const uint64_t cells = (uint64_t) 1 << (2 * (depth - blocks[key[k].second].level));
        if (key[k + 1].first < key[k].first + cells) { set_error("binary tree: blocks %d and %d overlap", key[k].second, key[k + 1].second); return MH_E_INVALID; }
    }
    for (int k = 0; k < nb; ++k) order[k] = key[k].second;
    return MH_OK;
}

} // namespace mh

extern "C" {

int mh_binary_tree_curve_order(const mh_tree_block* blocks, int nblocks, int32_t* order)
{
    if (! blocks || ! order || nblocks < 1) { mh::set_error("mh_binary_tree_curve_order: null argument"); return MH_E_INVALID; }
    return mh::binary_tree_curve_order(blocks, nblocks, order);
}

int mh_binary_tree_build(int bs, int depth, double focus_factor, double focus_index, mh_tree_block* out, int capacity)
{
    if (bs < 2 || bs % 2 != 0 || depth < 0 || depth > 12) return MH_E_INVALID;
    LeafSet T;
    T.leaves.insert({0, 0, 0});
    for (int it = 0; it < depth; ++it)
    {
        // bifurcate_if over every leaf with level = the loop counter (mesh_tree_operators.hpp:177-188); centroid of the block's
        // corner vertices, which are exact dyadic points of [-1, 1]
        std::vector<Key> todo;
        for (const auto& l : T.leaves)
        {
            const double w = 2.0 / (1 << l[0]);
            const double cx = ((-1.0 + w * l[1]) + (-1.0 + w * (l[1] + 1))) * 0.5, cy = ((-1.0 + w * l[2]) + (-1.0 + w * (l[2] + 1))) * 0.5;
            const double r = std::sqrt(0.0 + cx * cx + cy * cy);
            if (r < focus_factor / std::pow(double(it), focus_index)) todo.push_back(l);
        }
        for (const auto& k : todo) T.bifurcate(k);
    }
    for (;;)      // ensure_valid_quadtree :115-139
    {
        std::vector<Key> todo;
        for (const auto& l : T.leaves)
        {
            bool over = false;
            for (int s = 0; s < 4 && ! over; ++s)
            {
                Key n = l;
                n[1 + s / 2] += (s % 2) ? 1 : -1;
                over = T.node_depth(LeafSet::wrap(n)) > 1;
            }
            if (over) todo.push_back(l);
        }
        if (todo.empty()) break;
        for (const auto& k : todo) T.bifurcate(k);
    }
    std::vector<Key> order;
    T.ordered({0, 0, 0}, order);
    if (out && capacity >= (int) order.size())
        for (std::size_t n = 0; n < order.size(); ++n) out[n] = {order[n][0], order[n][1], order[n][2]};
    return (int) order.size();
}

int mh_binary_tree_vertices(int bs, double domain_radius, const mh_tree_block* blocks, int nblocks, double* edges)
{
    if (! blocks || ! edges || nblocks < 1) return MH_E_INVALID;
    std::map<int, std::vector<double>> cache;
    for (int b = 0; b < nblocks; ++b)
    {
        const int L = blocks[b].level;
        if (! cache.count(L)) cache[L] = level_vertices(bs, L);
        const auto& v = cache[L];
        double* e = edges + (std::size_t) b * 2 * (bs + 1);
        for (int a = 0; a <= bs; ++a)
        {
            e[a] = v[blocks[b].i * bs + a] * domain_radius;
            e[bs + 1 + a] = v[blocks[b].j * bs + a] * domain_radius;
        }
    }
    return MH_OK;
}

int mh_binary_tree_solver_data(const mh_binary_model* m, int bs, const mh_tree_block* blocks, int nblocks, const double* edges,
                               double* u_init, double* br, double* recommended_dt)
{
    if (! m || ! blocks || ! edges || ! u_init || ! br || ! recommended_dt || nblocks < 1) return MH_E_INVALID;
    double min_dx = 1e300, min_dy = 1e300, max_v = 1.0;        // std::max(make_velocity(1.0), ...) solver_data.cpp:53-58
    for (int b = 0; b < nblocks; ++b)
    {
        const double* xv = edges + (std::size_t) b * 2 * (bs + 1);
        double bx, by, bv;
        mh::binary_grid_data(m, bs, xv, xv + bs + 1, u_init + (std::size_t) b * bs * bs * 3, br + (std::size_t) b * bs * bs, &bx, &by, &bv);
        min_dx = std::fmin(min_dx, bx);
        min_dy = std::fmin(min_dy, by);
        max_v = std::fmax(max_v, bv);
    }
    *recommended_dt = std::fmin(min_dx, min_dy) / max_v * m->cfl_number;
    return MH_OK;
}

} // extern "C"
