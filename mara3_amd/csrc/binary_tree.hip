// `binary` on a GRADED block tree for gfx950 (the sub-program's default mesh).
//
// Leaf blocks of bs x bs cells at different levels. One evaluation of binary::advance_u (src/subprog_binary_scheme.cpp:790-904)
// runs as three kernels over the blocks and a one-workgroup reduction:
//   tree_prim_grad_kernel  p0    = recover_primitive(u)                                            :802
//                        gx, gy  = plm_gradient(p0 extended by one guard zone) / spacing(level)     :794-811
//   tree_flux_kernel     fhat_x, fhat_y on every face of every block, guard values of p0, gx, gy     :472-516
//   tree_update_kernel   flux correction at refinement jumps (:614-700), sources, update, totals     :568-587, :345-411
// Guard zones follow mara::get_cell_block (mesh_tree_operators.hpp:223-258): a neighbour of the same level is copied, a coarser
// one prolonged piecewise-constant (refine_cells), a finer one restricted - restrict_cells(0) then (1), i.e. pairs averaged
// along x first, then along y (mesh_prolong_restrict.hpp:124-132). A coarse face next to two fine blocks takes the sum of their
// two fluxes, even face first (restrict_extrinsic :134-142); fluxes carry their face length already.
//
// The per-cell and per-face arithmetic is the same policy code as the uniform kernel (binary.hip: BinStrict / BinFast; both conserved-variable forms), so on a
// uniform tree both kernel families produce identical bits (tests/test_gpu_binary_tree.py). These kernels are written for
// generality, not for the roofline: the graded runs of the reference are small (64 blocks of 24^2 cells at the defaults);
// primitives, slopes and fluxes make a round trip through memory between the kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "launch.hpp"
#include "binary_device.hpp"
#include "status_device.hpp"

namespace mh {

enum { NB_SAME = 0, NB_COARSER = 1, NB_FINER = 2 };

struct TreeGeom
{
    const int32_t* topo;      // [nb][4][3]: kind, id0, id1 (FINER: the two fine blocks in tangential order) or half (COARSER: which half of the coarse edge)
    const int32_t* level;     // [nb]
    const double*  edges;     // [nb][2][bs + 1]
    int nb, bs;
};

__device__ inline long cell_index(int bs, int b, int q, int i, int j) { return (((long) b * 3 + q) * bs + i) * bs + j; }

__device__ inline State3 load3(const double* F, int bs, int b, int i, int j)
{
    State3 s;
#pragma unroll
    for (int q = 0; q < 3; ++q) s[q] = F[cell_index(bs, b, q, i, j)];
    return s;
}

// value of a cell-centred quantity at (i, j) of block b, where ONE of i, j may be -1 or bs: the guard zone of get_cell_block.
// leaf(block, i, j) gives the quantity at an interior cell of a block (a load, or a recomputation)
template<class Leaf>
__device__ inline State3 fetch_with(const TreeGeom& g, int b, int i, int j, Leaf leaf)
{
    const int bs = g.bs;
    if (i >= 0 && i < bs && j >= 0 && j < bs) return leaf(b, i, j);
    const int side = i < 0 ? 0 : (i >= bs ? 1 : (j < 0 ? 2 : 3));
    const int axis = side >> 1;                    // 0: the neighbour lies along x
    const bool upper = side & 1;
    const int a = axis == 0 ? j : i;               // tangential index
    const int32_t* t = g.topo + ((long) b * 4 + side) * 3;
    const int kind = t[0];
    if (kind == NB_SAME)
    {
        const int n = upper ? 0 : bs - 1;
        return axis == 0 ? leaf(t[1], n, a) : leaf(t[1], a, n);
    }
    if (kind == NB_COARSER)
    {
        // refine_cells: the fine guard cell takes the value of the coarse cell it lies in
        const int n = upper ? 0 : bs - 1;
        const int tc = t[2] * (bs / 2) + a / 2;
        return axis == 0 ? leaf(t[1], n, tc) : leaf(t[1], tc, n);
    }
    // finer: coarsen_cells of the 2 x 2 fine cells under the guard cell: average along x, then along y
    const int fa = 2 * a;                          // first of the two fine tangential indices in the combined (2 bs) edge
    const int fb = fa >= bs ? t[2] : t[1];
    const int ta = fa >= bs ? fa - bs : fa;
    const int n0 = upper ? 0 : bs - 2;             // the two fine layers next to the face
    State3 r;
    if (axis == 0)
    {
        const State3 c00 = leaf(fb, n0, ta), c10 = leaf(fb, n0 + 1, ta), c01 = leaf(fb, n0, ta + 1), c11 = leaf(fb, n0 + 1, ta + 1);
#pragma unroll
        for (int q = 0; q < 3; ++q) r[q] = ((c00[q] + c10[q]) / 2 + (c01[q] + c11[q]) / 2) / 2;
    }
    else
    {
        const State3 c00 = leaf(fb, ta, n0), c10 = leaf(fb, ta + 1, n0), c01 = leaf(fb, ta, n0 + 1), c11 = leaf(fb, ta + 1, n0 + 1);
#pragma unroll
        for (int q = 0; q < 3; ++q) r[q] = ((c00[q] + c10[q]) / 2 + (c01[q] + c11[q]) / 2) / 2;
    }
    return r;
}

__device__ inline State3 fetch(const double* F, const TreeGeom& g, int b, int i, int j)
{
    const int bs = g.bs;
    return fetch_with(g, b, i, j, [F, bs] (int bb, int ii, int jj) { return load3(F, bs, bb, ii, jj); });
}

__device__ inline double spacing_of(const BinaryConsts& c, int level) { return c.h0 / (1 << level); }     // spacing_at_root / 2^level :793-799

// Primitives and slopes in one launch: a cell's primitives are stored (the flux and update kernels read them), those of its four
// neighbours are recovered again from the conserved field - same function, same inputs, same bits as the stored ones - so that the
// slopes need no earlier pass over the tree.
template<class A, bool QFORM>
__global__ __launch_bounds__(256)
void tree_prim_grad_kernel(const double* u, double* prim, double* gx, double* gy, TreeGeom g, BinaryConsts c, double theta, int b0)
{
    const int b = b0 + blockIdx.x, bs = g.bs;          // (b0: a member of a distributed tree runs its own blocks [b0, b1) only)
    BinaryConsts cb = c;
    cb.h = spacing_of(c, g.level[b]);
    const typename A::Ctx k = A::make(cb);
    const double* edges = g.edges;
    auto prim_of = [u, edges, bs] (int bb, int ii, int jj)
    {
        const double* xv = edges + (long) bb * 2 * (bs + 1);
        const double* yv = xv + bs + 1;
        return A::template c2p<QFORM>(load3(u, bs, bb, ii, jj), (xv[ii] + xv[ii + 1]) * 0.5, (yv[jj] + yv[jj + 1]) * 0.5);
    };
    for (int idx = blockIdx.y * 256 + threadIdx.x; idx < bs * bs; idx += 256 * gridDim.y)
    {
        const int i = idx / bs, j = idx - i * bs;
        const State3 P0 = prim_of(b, i, j);
        const State3 Gx = A::plm_per_length(fetch_with(g, b, i - 1, j, prim_of), P0, fetch_with(g, b, i + 1, j, prim_of), theta, k);
        const State3 Gy = A::plm_per_length(fetch_with(g, b, i, j - 1, prim_of), P0, fetch_with(g, b, i, j + 1, prim_of), theta, k);
#pragma unroll
        for (int q = 0; q < 3; ++q)
        {
            prim[cell_index(bs, b, q, i, j)] = P0[q];
            gx[cell_index(bs, b, q, i, j)] = Gx[q];
            gy[cell_index(bs, b, q, i, j)] = Gy[q];
        }
    }
}

// fx: [nb][3][bs + 1][bs], fy: [nb][3][bs][bs + 1]
__device__ inline long fx_index(int bs, int b, int q, int i, int j) { return (((long) b * 3 + q) * (bs + 1) + i) * bs + j; }
__device__ inline long fy_index(int bs, int b, int q, int i, int j) { return (((long) b * 3 + q) * bs + i) * (bs + 1) + j; }

template<class A, bool QFORM>
__global__ __launch_bounds__(256)
void tree_flux_kernel(const double* prim, const double* gx, const double* gy, double* fx, double* fy, TreeGeom g, BinaryConsts c, int b0)
{
    const int b = b0 + blockIdx.x, bs = g.bs;
    BinaryConsts cb = c;
    cb.h = spacing_of(c, g.level[b]);
    const typename A::Ctx k = A::make(cb);
    const double* xv = g.edges + (long) b * 2 * (bs + 1);
    const double* yv = xv + bs + 1;
    // grid = (blocks, 2 x tiles of 256 faces): even blockIdx.y -> x-faces, odd -> y-faces. A 64-block tree would otherwise occupy 64 of
    // the 256 CUs with three sequential passes of two face fluxes each.
    const bool xfaces = (blockIdx.y & 1) == 0;
    for (int idx = (blockIdx.y >> 1) * 256 + threadIdx.x; idx < (bs + 1) * bs; idx += 256 * (gridDim.y >> 1))
    {
        if (xfaces)
        {   // x-face (i, j), i = 0..bs
            const int i = idx / bs, j = idx - i * bs;
            const double xf = (xv[i] + xv[i]) * 0.5, yf = (yv[j] + yv[j + 1]) * 0.5;
            State3 F = binary_face_flux<A, 0, QFORM>(cb, k, xf, yf, fetch(prim, g, b, i - 1, j), fetch(prim, g, b, i, j),
                                                     fetch(gx, g, b, i - 1, j), fetch(gx, g, b, i, j), fetch(gy, g, b, i - 1, j), fetch(gy, g, b, i, j));
            const double dy = yv[j + 1] - yv[j];
#pragma unroll
            for (int q = 0; q < 3; ++q) fx[fx_index(bs, b, q, i, j)] = F[q] * dy;
        }
        else
        {   // y-face (i, j), j = 0..bs
            const int i = idx / (bs + 1), j = idx - i * (bs + 1);
            const double xf = (xv[i] + xv[i + 1]) * 0.5, yf = (yv[j] + yv[j]) * 0.5;
            State3 F = binary_face_flux<A, 1, QFORM>(cb, k, xf, yf, fetch(prim, g, b, i, j - 1), fetch(prim, g, b, i, j),
                                                     fetch(gy, g, b, i, j - 1), fetch(gy, g, b, i, j), fetch(gx, g, b, i, j - 1), fetch(gx, g, b, i, j));
            const double dx = xv[i + 1] - xv[i];
#pragma unroll
            for (int q = 0; q < 3; ++q) fy[fy_index(bs, b, q, i, j)] = F[q] * dx;
        }
    }
}

// the flux through x-face i (0 or bs on the block's edge, anything inside) of row j after correct_fluxes_x
__device__ inline double corrected_fx(const double* fx, const TreeGeom& g, int b, int q, int i, int j)
{
    const int bs = g.bs;
    if (i == 0 || i == bs)
    {
        const int32_t* t = g.topo + ((long) b * 4 + (i == 0 ? 0 : 1)) * 3;
        if (t[0] == NB_FINER)
        {
            const int fa = 2 * j, fb = fa >= bs ? t[2] : t[1], ta = fa >= bs ? fa - bs : fa;
            const int fi = i == 0 ? bs : 0;                       // the fine blocks' face that coincides with ours
            return fx[fx_index(bs, fb, q, fi, ta)] + fx[fx_index(bs, fb, q, fi, ta + 1)];
        }
    }
    return fx[fx_index(bs, b, q, i, j)];
}
__device__ inline double corrected_fy(const double* fy, const TreeGeom& g, int b, int q, int i, int j)
{
    const int bs = g.bs;
    if (j == 0 || j == bs)
    {
        const int32_t* t = g.topo + ((long) b * 4 + (j == 0 ? 2 : 3)) * 3;
        if (t[0] == NB_FINER)
        {
            const int fa = 2 * i, fb = fa >= bs ? t[2] : t[1], ta = fa >= bs ? fa - bs : fa;
            const int fj = j == 0 ? bs : 0;
            return fy[fy_index(bs, fb, q, ta, fj)] + fy[fy_index(bs, fb, q, ta + 1, fj)];
        }
    }
    return fy[fy_index(bs, b, q, i, j)];
}

static constexpr int NTREE_SUMS = 16;      // per block: mass_acc[2], L_acc[2], torque[2], px_acc[2], py_acc[2], fx[2], fy[2], mass_ej, L_ej

template<class A, bool COMBINE, bool QFORM, bool MAXW>
__global__ __launch_bounds__(256)
void tree_update_kernel(const double* u_in, const double* u_base, double* u_out, const double* u_init, const double* br, const double* prim,
                        const double* fx, const double* fy, TreeGeom g, BinaryConsts c, BinaryConsts c_next, double dt, double weight, double* block_out,
                        double* tile_maxw, int32_t* status, int b0, const int32_t* ids)
{
    // ids (or null): the caller's number of the block stored at each position (a distributed tree stores its blocks in curve order)
    const int b = b0 + blockIdx.x, bs = g.bs;
    const uint32_t b_id = ids ? (uint32_t) ids[b] : (uint32_t) b;
    double wmax = 0.0;        // MAXW: as tree_maxw_kernel, on the state this launch writes, with the bodies of the next step's start
    const double* xv = g.edges + (long) b * 2 * (bs + 1);
    const double* yv = xv + bs + 1;
    double acc[NTREE_SUMS];
#pragma unroll
    for (int k = 0; k < NTREE_SUMS; ++k) acc[k] = 0.0;
    StatusAcc sacc;       // validate_u as status bits + first failing cell (block * bs + i) * bs + j, the order of mh_binary_get_solution
    // grid = (blocks, tiles of 256 cells): one cell per thread, a partial sum per tile (a 64-block tree would otherwise run on 64 CUs)
    for (int idx = blockIdx.y * 256 + threadIdx.x; idx < bs * bs; idx += 256 * gridDim.y)
    {
        const int i = idx / bs, j = idx - i * bs;
        const double xc = (xv[i] + xv[i + 1]) * 0.5, yc = (yv[j] + yv[j + 1]) * 0.5;
        const double dA = (xv[i + 1] - xv[i]) * (yv[j + 1] - yv[j]);
        const State3 u0 = load3(u_in, bs, b, i, j);
        double fg[2][2], s_grav[2][3], s_sink[2][3], s_buffer[3], s_floor[3];
#pragma unroll
        for (int bdy = 0; bdy < 2; ++bdy)
        {
            const double d0 = xc - c.body[5 * bdy + 1], d1 = yc - c.body[5 * bdy + 2];
            A::gravity(c, bdy, d0, d1, u0[0], fg[bdy]);
            s_grav[bdy][0] = 0.0 * dt;
            s_grav[bdy][1] = fg[bdy][0] * dt;
            s_grav[bdy][2] = fg[bdy][1] * dt;
            const double a2 = A::sink_a2(c, d0, d1);
            const double rate = c.sink_rate * (a2 < 750.0 ? exp(-a2) : 0.0);
#pragma unroll
            for (int q = 0; q < 3; ++q) s_sink[bdy][q] = -u0[q] * rate * dt;
        }
        const double fl = u0[0] < c.floor_sigma ? 1.0 : 0.0;
        const double brate = br[((long) b * bs + i) * bs + j];
#pragma unroll
        for (int q = 0; q < 3; ++q)
        {
            s_buffer[q] = (u_init[cell_index(bs, b, q, i, j)] - u0[q]) * brate * dt;
            s_floor[q] = u0[q] * 1e-2 * fl;
        }
        double dps[2][2] = {{s_sink[0][1], s_sink[0][2]}, {s_sink[1][1], s_sink[1][2]}};
        if constexpr (QFORM)
        {
            // source_terms_q :417-466 (as in binary.hip)
            const typename A::Ctx k = A::make(c);
            const double r2 = 0.0 + xc * xc + yc * yc;
#pragma unroll
            for (int bdy = 0; bdy < 2; ++bdy)
            {
                s_grav[bdy][1] = (xc * fg[bdy][0] + yc * fg[bdy][1]) * dt;
                s_grav[bdy][2] = (xc * fg[bdy][1] - yc * fg[bdy][0]) * dt;
                dps[bdy][0] = (s_sink[bdy][1] * xc - s_sink[bdy][2] * yc) / r2;
                dps[bdy][1] = (s_sink[bdy][1] * yc + s_sink[bdy][2] * xc) / r2;
            }
            const double a = -(xc * xc + yc * yc) / c.sr2;
            const double ramp = 1.0 - (a > -750.0 ? exp(a) : 0.0);
            const State3 pc = load3(prim, bs, b, i, j);
            const double Ek = 0.5 * pc[0] * (pc[1] * pc[1] + pc[2] * pc[2]);
            const double pg = pc[0] * A::cs2(c, k, xc, yc);
            s_floor[0] = 0.0 * ramp * dt;
            s_floor[1] = (Ek + pg) * 2.0 * ramp * dt;
            s_floor[2] = 0.0 * ramp * dt;
        }
#pragma unroll
        for (int bdy = 0; bdy < 2; ++bdy)
        {
            acc[0 + bdy]  = acc[0 + bdy] + s_sink[bdy][0] * dA;
            acc[2 + bdy]  = acc[2 + bdy] + (QFORM ? s_sink[bdy][2] : (xc * s_sink[bdy][2] - yc * s_sink[bdy][1])) * dA;
            acc[4 + bdy]  = acc[4 + bdy] + (QFORM ? s_grav[bdy][2] : (xc * s_grav[bdy][2] - yc * s_grav[bdy][1])) * dA;
            acc[6 + bdy]  = acc[6 + bdy] + dps[bdy][0] * dA;
            acc[8 + bdy]  = acc[8 + bdy] + dps[bdy][1] * dA;
            acc[10 + bdy] = acc[10 + bdy] + fg[bdy][0] * dt * dA;
            acc[12 + bdy] = acc[12 + bdy] + fg[bdy][1] * dt * dA;
        }
        acc[14] = acc[14] + s_buffer[0] * dA;
        acc[15] = acc[15] + (QFORM ? s_buffer[2] : (xc * s_buffer[2] - yc * s_buffer[1])) * dA;

        double l[3];
        State3 unew;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            l[q] = ((corrected_fx(fx, g, b, q, i + 1, j) - corrected_fx(fx, g, b, q, i, j)) + (corrected_fy(fy, g, b, q, i, j + 1) - corrected_fy(fy, g, b, q, i, j))) * dt;
        A::over_area(l, dA);
#pragma unroll
        for (int q = 0; q < 3; ++q)
        {
            const double s = s_grav[0][q] + s_grav[1][q] + s_sink[0][q] + s_sink[1][q] + s_buffer[q] + s_floor[q];
            const double u1 = u0[q] - l[q] + s;
            if (q == 0 && !(u1 >= 0.0)) sacc.note_value(u1, MH_STATUS_NEG_DENSITY, (b_id * (uint32_t) bs + (uint32_t) i) * (uint32_t) bs + (uint32_t) j);
            double un = u1;
            if constexpr (COMBINE) un = u_base[cell_index(bs, b, q, i, j)] * (1.0 - weight) + u1 * weight;
            u_out[cell_index(bs, b, q, i, j)] = un;
            unew[q] = un;
        }
        if constexpr (MAXW)
        {
            State3 Pn;
            if constexpr (QFORM) iso2d::recover_primitive_angmom(unew, xc, yc, Pn);
            else                 iso2d::recover_primitive(unew, Pn);
            const double w = iso2d::max_wavespeed(Pn, binary_cs2(c_next, make_recip(c_next.mach, 1.0), xc, yc));
            wmax = (wmax < w) ? w : wmax;
        }
    }
    __shared__ double red[NTREE_SUMS][256];
#pragma unroll
    for (int k = 0; k < NTREE_SUMS; ++k) red[k][threadIdx.x] = acc[k];
    __shared__ double wred[256];
    if constexpr (MAXW) wred[threadIdx.x] = wmax;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1)
    {
        if ((int) threadIdx.x < off)
        {
            for (int k = 0; k < NTREE_SUMS; ++k) red[k][threadIdx.x] = red[k][threadIdx.x] + red[k][threadIdx.x + off];
            if constexpr (MAXW) wred[threadIdx.x] = wred[threadIdx.x] < wred[threadIdx.x + off] ? wred[threadIdx.x + off] : wred[threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        double* out = block_out + ((long) b * gridDim.y + blockIdx.y) * NTREE_SUMS;       // raw partial sums of this tile; signs and work: tree_totals_kernel
        for (int k = 0; k < NTREE_SUMS; ++k) out[k] = red[k][0];
        if constexpr (MAXW) tile_maxw[(long) b * gridDim.y + blockIdx.y] = wred[0];        // largest wavespeed of the tile's new state
    }
    sacc.commit(status);
}

// totals[18] in the order of mh_binary_total (mass_acc, L_acc, torque, px_acc, py_acc, fx, fy, work, mass_ej, L_ej): per block the tile
// partials are added in tile order, the block values in the tree's traversal order (scheme.cpp:829-830 sums the blocks' results
// with tree.sum()); work_done_on :356-365 is a per-block function of that block's sink sums.
// One workgroup. Phase 1: all threads form the nb x 18 block values (independent loads, into block_vals); phase 2: one thread per
// entry adds them in block order, the loads of eight blocks issued together - a plain loop is a chain of dependent load latencies
// (13 us for 64 blocks, 100 us with the tile sums inside it).
__global__ __launch_bounds__(1024)
void tree_totals_kernel(const double* partial, int nb, int tiles, BinaryConsts c, int qform, double* block_vals, double* totals,
                        const double* tile_maxw, const int32_t* level, double* maxw_result, const int32_t* order)
{
    // order (or null): where the caller's k-th block is stored; the block values are added in the CALLER's order, so that a tree stored in
    // curve order (distributed) forms the same sums, bit for bit, as the tree stored as given
    auto block_sum = [&] (int b, int k)
    {
        double v = 0.0;
        for (int tile = 0; tile < tiles; ++tile) v = v + partial[((long) b * tiles + tile) * NTREE_SUMS + k];
        return -v;
    };
    for (int idx = threadIdx.x; idx < nb * MH_BINARY_NTOTALS; idx += blockDim.x)
    {
        const int b = idx / MH_BINARY_NTOTALS, t = idx - b * MH_BINARY_NTOTALS;
        double val;
        if (t < MH_T_WORK) val = block_sum(b, t);
        else if (t >= MH_T_MASS_EJ) val = block_sum(b, 14 + (t - MH_T_MASS_EJ));
        else if (qform) val = 0.0;
        else
        {
            const int bdy = t - MH_T_WORK;
            const double M0 = c.body[5 * bdy], px0 = c.body[5 * bdy + 3] * M0, py0 = c.body[5 * bdy + 4] * M0;
            const double M1 = M0 + block_sum(b, MH_T_MASS_ACC + bdy), px1 = px0 + block_sum(b, MH_T_PX_ACC + bdy), py1 = py0 + block_sum(b, MH_T_PY_ACC + bdy);
            val = ((px1 * px1 + py1 * py1) / M1 - (px0 * px0 + py0 * py0) / M0) * 0.5;
        }
        block_vals[idx] = val;
    }
    // after a step's final stage: the next step's time-step bound, min over blocks of spacing / max over the block's tiles
    // (maximum_timestep :1107-1126; max and min are order-independent, so this equals tree_maxw_kernel's result on the same state)
    __shared__ double dred[1024];
    if (tile_maxw)
    {
        double dtmin = __longlong_as_double(0x7ff0000000000000LL);
        for (int b = threadIdx.x; b < nb; b += blockDim.x)
        {
            double m = 0.0;
            for (int tile = 0; tile < tiles; ++tile) { const double w = tile_maxw[(long) b * tiles + tile]; m = (m < w) ? w : m; }
            const double dtb = spacing_of(c, level[b]) / m;
            dtmin = dtb < dtmin ? dtb : dtmin;
        }
        dred[threadIdx.x] = dtmin;
    }
    __threadfence_block();
    __syncthreads();
    if (tile_maxw)
    {
        for (int off = 512; off > 0; off >>= 1)
        {
            if ((int) threadIdx.x < off) dred[threadIdx.x] = dred[threadIdx.x + off] < dred[threadIdx.x] ? dred[threadIdx.x + off] : dred[threadIdx.x];
            __syncthreads();
        }
        if (threadIdx.x == 0) *maxw_result = dred[0];
    }
    const int t = threadIdx.x;
    if (t >= MH_BINARY_NTOTALS) return;
    double s = 0.0;
    int b = 0;
    for (; b + 8 <= nb; b += 8)
    {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = block_vals[(long) (order ? order[b + k] : b + k) * MH_BINARY_NTOTALS + t];
#pragma unroll
        for (int k = 0; k < 8; ++k) s = s + v[k];
    }
    for (; b < nb; ++b) s = s + block_vals[(long) (order ? order[b] : b) * MH_BINARY_NTOTALS + t];
    totals[t] = s;
}

// min over blocks of spacing / max over its cells of max_wavespeed (maximum_timestep :1107-1126); *result must hold +inf beforehand
__global__ __launch_bounds__(256)
void tree_maxw_kernel(const double* u, TreeGeom g, BinaryConsts c, int qform, unsigned long long* result)
{
    const int b = blockIdx.x, bs = g.bs;
    const Recip rmach = make_recip(c.mach, 1.0);
    const double* xv = g.edges + (long) b * 2 * (bs + 1);
    const double* yv = xv + bs + 1;
    double m = 0.0;
    for (int idx = threadIdx.x; idx < bs * bs; idx += 256)
    {
        const int i = idx / bs, j = idx - i * bs;
        State3 P;
        const double xc = (xv[i] + xv[i + 1]) * 0.5, yc = (yv[j] + yv[j + 1]) * 0.5;
        if (qform) iso2d::recover_primitive_angmom(load3(u, bs, b, i, j), xc, yc, P);
        else       iso2d::recover_primitive(load3(u, bs, b, i, j), P);
        const double w = iso2d::max_wavespeed(P, binary_cs2(c, rmach, xc, yc));
        m = (m < w) ? w : m;
    }
    __shared__ double red[256];
    red[threadIdx.x] = m;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1)
    {
        if ((int) threadIdx.x < off) red[threadIdx.x] = red[threadIdx.x] < red[threadIdx.x + off] ? red[threadIdx.x + off] : red[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        const double dtb = spacing_of(c, g.level[b]) / red[0];
        atomicMin(result, (unsigned long long) __double_as_longlong(dtb));      // positive doubles order as integers
    }
}

__global__ void tree_set_inf_kernel(double* x) { *x = __longlong_as_double(0x7ff0000000000000LL); }

// ---- launchers ---------------------------------------------------------------------------------------------------------------
BinaryConsts binary_make_consts(const mh_binary_desc* d, const double bodies[10]);

struct TreeBuffers { double *prim, *gx, *gy, *fx, *fy, *block_out, *block_vals, *tile_maxw; };

// What one call of binary_tree_stage_launch runs. A member of a DISTRIBUTED tree (binary_api.hip) runs the three block kernels on its own
// blocks [b0, b1) only, one call per kernel with the other members' results gathered in between, and then the totals over all blocks.
enum { TREE_PRIM_GRAD = 1, TREE_FLUX = 2, TREE_UPDATE = 4, TREE_TOTALS = 8, TREE_ALL = 15 };
struct TreeRun { int phases, b0, b1; const int32_t* order; const int32_t* ids; };          // order, ids: tree_totals_kernel / tree_update_kernel

// bodies_next != nullptr: the stage also leaves min over blocks of spacing / largest wavespeed of the state it writes, evaluated with
// those bodies, in *maxw_result (what binary_tree_min_dt_launch computes in two launches of its own)
hipError_t binary_tree_stage_launch(const mh_binary_desc* d, const TreeGeom& g, const TreeBuffers& w, const double* u_in, const double* u_base,
                                    double* u_out, const double* u_init, const double* br, const double bodies[10], double dt, double weight,
                                    double theta, double* totals, int32_t* status, hipStream_t stream, const double* bodies_next, double* maxw_result,
                                    const TreeRun* part)
{
    BinaryConsts c = binary_make_consts(d, bodies);
    binary_set_theta(c, theta);            // the STAGE's theta (safe mode: 0)
    BinaryConsts cn = bodies_next ? binary_make_consts(d, bodies_next) : c;
    const dim3 blk(256);
    const bool fast = d->arith == MH_ARITH_FAST, combine = weight != 1.0, q = d->angmom_form != 0, maxw = bodies_next != nullptr;
    const int phases = part ? part->phases : TREE_ALL, b0 = part ? part->b0 : 0, b1 = part ? part->b1 : g.nb;
    const int32_t* ids = part ? part->ids : nullptr;
    if (b0 < 0 || b1 > g.nb || b0 > b1) return hipErrorInvalidValue;
    auto run = [&] (auto policy, auto qform)
    {
        using A = decltype(policy);
        constexpr bool Q = decltype(qform)::value;
        const unsigned cell_tiles = (unsigned) ((g.bs * g.bs + 255) / 256), face_tiles = (unsigned) (((g.bs + 1) * g.bs + 255) / 256);
        const unsigned own = (unsigned) (b1 - b0);
        if (own == 0) return;
        if (phases & TREE_PRIM_GRAD) hipLaunchKernelGGL((tree_prim_grad_kernel<A, Q>), dim3(own, cell_tiles), blk, 0, stream, u_in, w.prim, w.gx, w.gy, g, c, theta, b0);
        if (phases & TREE_FLUX) hipLaunchKernelGGL((tree_flux_kernel<A, Q>), dim3(own, 2 * face_tiles), blk, 0, stream, w.prim, w.gx, w.gy, w.fx, w.fy, g, c, b0);
        if (! (phases & TREE_UPDATE)) return;
        const dim3 ugrid(own, cell_tiles);
#define MH_TREE_UPDATE(COMBINE, MAXW) hipLaunchKernelGGL((tree_update_kernel<A, COMBINE, Q, MAXW>), ugrid, blk, 0, stream, u_in, u_base, u_out, u_init, br, \
                                                         w.prim, w.fx, w.fy, g, c, cn, dt, weight, w.block_out, w.tile_maxw, status, b0, ids)
        if (combine) { if (maxw) MH_TREE_UPDATE(true, true); else MH_TREE_UPDATE(true, false); }
        else         { if (maxw) MH_TREE_UPDATE(false, true); else MH_TREE_UPDATE(false, false); }
#undef MH_TREE_UPDATE
    };
    if (fast) { if (q) run(BinFast(), std::true_type()); else run(BinFast(), std::false_type()); }
    else      { if (q) run(BinStrict(), std::true_type()); else run(BinStrict(), std::false_type()); }
    if (phases & TREE_TOTALS)
        hipLaunchKernelGGL(tree_totals_kernel, dim3(1), dim3(1024), 0, stream, w.block_out, g.nb, (g.bs * g.bs + 255) / 256, c, (int) q, w.block_vals, totals,
                           maxw ? w.tile_maxw : nullptr, g.level, maxw_result, part ? part->order : nullptr);
    return hipGetLastError();
}

hipError_t binary_tree_min_dt_launch(const mh_binary_desc* d, const TreeGeom& g, const double* u, const double bodies[10], double* result, hipStream_t stream)
{
    const BinaryConsts c = binary_make_consts(d, bodies);
    hipLaunchKernelGGL(tree_set_inf_kernel, dim3(1), dim3(1), 0, stream, result);
    hipLaunchKernelGGL(tree_maxw_kernel, dim3(g.nb), dim3(256), 0, stream, u, g, c, (int) d->angmom_form, reinterpret_cast<unsigned long long*>(result));
    return hipGetLastError();
}

} // namespace mh
