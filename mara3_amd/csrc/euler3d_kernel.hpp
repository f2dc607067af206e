// The 3-D Euler stage kernel and its launcher template (see euler3d.hip for the design). Shared by two translation units so that the
// STRICT and the FAST instantiations can be compiled with different scheduling strategies (Makefile: euler3d_fast.o).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "euler_device.hpp"
#include "euler_device_fast.hpp"
#include <atomic>
#include "launch.hpp"
#include "status_device.hpp"
#include "row_check.hpp"

namespace mh {

static constexpr int W3 = 64;                  // lanes
static constexpr int H3 = 2;                   // halo
static constexpr int STRIP3 = W3 - 2 * H3;     // 60 output columns per wave
// axis-1 rows per workgroup (= waves): A::tile_rows (euler_device_fast.hpp) - 8: one workgroup per CU; 4: two independent workgroups per CU
static constexpr int MAX_BOXES = 8;

struct Stage3dParams
{
    const double* u_in;
    const double* u_base;
    double*       u_out;
    int32_t*      status;
    long   plane_stride;     // (n1 + 2 g1) * (n2 + 2 g2): doubles between variables of one axis-0 plane
    long   row_stride;       // 5 * plane_stride: doubles between consecutive axis-0 planes
    long   pitch2;           // n2 + 2 g2: doubles between consecutive axis-1 rows of a plane
    int    n0, n1, n2;       // cells of this field (without ghosts)
    int    g1, g2;           // stored ghost layers on axes 1 and 2 (0, or 2 for a block of a 3-axis decomposition)
    int    chunk_rows;
    int    nboxes;           // the launch covers up to MAX_BOXES boxes of (axis-0 rows) x (axis-1 tiles) x (axis-2 strips)
    Euler3dBox box[MAX_BOXES];
    int    first_block[MAX_BOXES + 1];       // workgroups [first_block[k], first_block[k + 1]) work on box k
    int    bc_lo0, bc_hi0, bc_lo1, bc_hi1, bc_lo2, bc_hi2;
    double gamma, theta, cx, cy, cz, weight;
};

__device__ inline double dpp3_left(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double dpp3_right(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline State5 dpp3_left(const State5& s) { State5 r; for (int q = 0; q < 5; ++q) r[q] = dpp3_left(s[q]); return r; }
__device__ inline State5 dpp3_right(const State5& s) { State5 r; for (int q = 0; q < 5; ++q) r[q] = dpp3_right(s[q]); return r; }

// Transverse index with the boundary condition of its side folded in: outflow clamps to the edge cell, periodic wraps, and an
// EXTERNAL side (a cut of the block decomposition) keeps the index, which then addresses the stored ghost cells [-2, -1] / [n, n + 1]
// that the neighbour's face filled. Lanes far outside (unused halo lanes of the last strip) are clamped into the stored range.
__device__ inline int fold_index(int j, int n, int bc_lo, int bc_hi)
{
    if (j < 0)  j = bc_lo == MH_BC_EXTERNAL ? j : (bc_lo == MH_BC_PERIODIC ? j + n : 0);
    if (j >= n) j = bc_hi == MH_BC_EXTERNAL ? j : (bc_hi == MH_BC_PERIODIC ? j - n : n - 1);
    return min(max(j, bc_lo == MH_BC_EXTERNAL ? -H3 : 0), bc_hi == MH_BC_EXTERNAL ? n - 1 + H3 : n - 1);
}

// LDS exchange buffer: primitives of one axis-0 plane for the tile's rows and two rows on either side, double-buffered by
// the parity of the plane index: [parity][slot = tile row + 2][variable][lane]
template<int ROWS3>
struct Tile3dT
{
    double P[2][ROWS3 + 2 * H3][5][W3];
    double U[ROWS3][3][5][W3];             // per-wave private ring: conserved state of planes r, r+1, r+2 (no barrier needed)
    double F[2][ROWS3][5][W3];             // deferred axis-1 mode only (last member: the other kernels do not allocate it): the flux through the
                                           // UPPER axis-1 face of each tile row, by plane parity - slot j is written by wave j + 1 (its lower
                                           // face), slot ROWS3 - 1 by the top wave itself
};
template<class A> constexpr size_t tile3d_bytes()
{
    using T = Tile3dT<A::tile_rows>;
    return A::deferred_axis1 ? sizeof(T) : sizeof(T) - sizeof(T::F);
}

__device__ inline void lds_put(double (*dst)[W3], int lane, const State5& s)
{
#pragma unroll
    for (int q = 0; q < 5; ++q) dst[q][lane] = s[q];
}
__device__ inline State5 lds_get(double (*src)[W3], int lane)
{
    State5 s;
#pragma unroll
    for (int q = 0; q < 5; ++q) s[q] = src[q][lane];
    return s;
}

using b64x_t = decltype(__builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t(), 0, 0, 0));

// the 5 variables of one cell per lane out of one axis-0 plane: wave-uniform plane pointer (scalar), per-lane byte offset of
// the cell, scalar offset of the variable (cdna_hip_programming.md T8). The descriptor spans exactly the plane's 5 variables.
__device__ inline State5 load_plane(const double* plane_ptr, long plane_doubles, unsigned cell_bytes)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(plane_ptr), 0, (int) (5 * plane_doubles * 8), 0x00020000);
    State5 U;
#pragma unroll
    for (int q = 0; q < 5; ++q)
        U[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, cell_bytes, (unsigned) (q * plane_doubles * 8), 0));
    return U;
}
__device__ inline void store_plane(double* plane_ptr, long plane_doubles, unsigned cell_bytes, const State5& U)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(plane_ptr, 0, (int) (5 * plane_doubles * 8), 0x00020000);
#pragma unroll
    for (int q = 0; q < 5; ++q)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(b64x_t, U[q]), rs, cell_bytes, (unsigned) (q * plane_doubles * 8), 0);
}

template<class A, int RIEMANN, bool PLM, bool COMBINE>
__global__ __launch_bounds__(W3 * A::tile_rows, 2)
void euler3d_stage_kernel(Stage3dParams p)
{
    extern __shared__ double lds_raw[];
    constexpr int ROWS3 = A::tile_rows;
    Tile3dT<ROWS3>& tile = *reinterpret_cast<Tile3dT<ROWS3>*>(lds_raw);

    // work item -> (chunk along axis 0, tile along axis 1, strip along axis 2); neighbouring items share an XCD
    int b = blockIdx.x;
    {
        const int per_xcd = gridDim.x >> 3;
        if (b < per_xcd * 8) b = (b & 7) * per_xcd + (b >> 3);
    }
    // which box of the launch (a single box for a whole-field stage; the boundary shell of a block is several)
    int bk = 0;
    while (bk + 1 < p.nboxes && b >= p.first_block[bk + 1]) ++bk;
    const Euler3dBox bx = p.box[bk];
    const int lb = b - p.first_block[bk];
    const int box_strips = bx.s1 - bx.s0, box_tiles = bx.t1 - bx.t0;
    const int strip = bx.s0 + lb % box_strips;
    const int t1 = bx.t0 + (lb / box_strips) % box_tiles;
    const int chunk = lb / (box_strips * box_tiles);
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));   // wave index = tile row (uniform -> scalar registers)
    const int r0 = bx.r0 + chunk * p.chunk_rows;
    const int r1 = min(r0 + p.chunk_rows, bx.r1);

    const int j = t1 * ROWS3 + row;                         // axis-1 index of this wave's row (may exceed n1 - 1 in the last tile)
    const int jc = fold_index(j, p.n1, p.bc_lo1, p.bc_hi1);
    const int col = strip * STRIP3 - H3 + lane;
    const int kc = fold_index(col, p.n2, p.bc_lo2, p.bc_hi2);
    const bool writes = lane >= H3 && lane < W3 - H3 && col < p.n2 && j < p.n1;

    const long row_stride = p.row_stride, plane = p.plane_stride;
    const int planes_hi = p.n0 + 1;                   // the planes that exist: -2 .. n0 + 1 (row_check.hpp)
    auto row_off = [row_stride, planes_hi] (int r) { (void) planes_hi; return (long) (MH_ROW(r, -H3, planes_hi) + H3) * row_stride; };
    auto cell_bytes = [&p] (int jj, int kk) { return (unsigned) (((long) (jj + p.g1) * p.pitch2 + (kk + p.g2)) * 8); };
    const unsigned c0 = cell_bytes(jc, kc);
    const unsigned cw = writes ? cell_bytes(j, col) : 0u;

    // Rows just outside the tile (two on either side with PLM, one without): the first four waves each fetch one of them
    // per plane and publish its primitives, so that this duty is spread instead of loading the tile's edge waves.
    //   wave 0: row -2   wave 1: row -1   wave 2: row ROWS3   wave 3: row ROWS3 + 1
    // (deferred axis-1 mode: waves 0, 1, 2 and 4 - the top wave 7 solves the tile's ninth face, and wave 3 shares its SIMD)
    constexpr bool DEFER = A::deferred_axis1;
    static_assert(! DEFER || ! PLM || A::shared_differences, "the deferred axis-1 mode is written for the shared-difference limiter");
    static_assert(ROWS3 == 8 || (ROWS3 == 4 && ! DEFER), "tiles of eight rows, or of four without the deferred axis-1 faces");
    const int hrow = DEFER ? (row == 3 ? 4 : (row == 4 ? 3 : row)) : row;          // (four-row tiles: every wave fetches one outside row)
    const int eoff = hrow == 0 ? -2 : (hrow == 1 ? -1 : (hrow == 2 ? ROWS3 : ROWS3 + 1));
    const bool helper = hrow < 4 && (PLM || hrow == 1 || hrow == 2);
    const unsigned ce = cell_bytes(fold_index(t1 * ROWS3 + eoff, p.n1, p.bc_lo1, p.bc_hi1), kc);
    const int eslot = eoff + H3;

    const double theta = p.theta;
    const typename A::Gamma gl = A::gamma_law(p.gamma);
    const typename A::Limiter lim = A::limiter(theta);
    const double* in = p.u_in;

    // ---- register window along axis 0: three slots used as rings (index = plane mod 3 relative to the chunk start)
    // The conserved state of a plane is needed twice, three planes apart (primitives, then the update). Between the two uses
    // it waits in a private LDS ring instead of 20 VGPRs: this kernel sits at the 256-register limit, and reading the plane
    // again from memory missed in L2 (FETCH_SIZE doubled).
    State5 P[3], G[3], Fx[3];
    {
        const State5 Pa = A::c2p(load_plane(in + row_off(r0 - 2), plane, c0), gl);
        const State5 Pb = A::c2p(load_plane(in + row_off(r0 - 1), plane, c0), gl);
        const State5 Ua = load_plane(in + row_off(r0), plane, c0), Ub = load_plane(in + row_off(r0 + 1), plane, c0);
        lds_put(tile.U[row][0], lane, Ua);
        lds_put(tile.U[row][1], lane, Ub);
        P[0] = A::c2p(Ua, gl);
        P[1] = A::c2p(Ub, gl);
        if constexpr (PLM)
        {
            const State5 Gb = A::plm(Pa, Pb, P[0], lim);
            G[0] = A::plm(Pb, P[0], P[1], lim);
            Fx[0] = A::template flux<RIEMANN, 0>(A::plus(Pb, Gb, lim), A::minus(P[0], G[0], lim), gl);
        }
        else
        {
            Fx[0] = A::template flux<RIEMANN, 0>(Pb, P[0], gl);
        }
    }
    State5 Uin = load_plane(in + row_off(r0 + 2), plane, c0);      // plane r+2, in flight for one iteration
    // error contract as in euler2d.hip: kind + flat cell index (r * n1 + j) * n2 + col behind wave-wide votes
    StatusAcc acc;
    const uint32_t cellu = (uint32_t) j * (uint32_t) p.n2 + (uint32_t) col, planeu = (uint32_t) p.n1 * (uint32_t) p.n2;
    if (__any(writes && (!(P[0][4] >= 0.0) || !(P[1][4] >= 0.0))))      // (halo lanes may hold never-used corner ghosts: they do not vote)
    {
        if (writes && !(P[0][4] >= 0.0)) acc.note_value(P[0][4], MH_STATUS_NEG_PRESSURE, (uint32_t) r0 * planeu + cellu);
        if (writes && !(P[1][4] >= 0.0) && r0 + 1 < p.n0) acc.note_value(P[1][4], MH_STATUS_NEG_PRESSURE, (uint32_t) (r0 + 1) * planeu + cellu);
    }

    // ---- deferred axis-1 mode (MH_ARITH_FAST): ONE Riemann problem per axis-1 face. Every wave solves the LOWER face of its row and leaves the
    // flux in LDS for the wave below it, whose update of that plane then waits one plane step for it: the plane's update is formed up to the
    // upper-face term (`pending`), and completed behind the NEXT plane's barrier, which makes the neighbour's flux visible. The top wave also
    // solves the tile's ninth face (it shares its SIMD with wave 3, which has no helper row: the four SIMDs carry 2 x 3 + 1 ... solves).
    // Same fluxes as the redundant form (it solved each interior face twice on identical inputs); the update adds its terms in another
    // order - x, z, lower y, upper y - so results differ from that form by rounding, inside MH_ARITH_FAST's tolerance; conservation holds to
    // rounding as before (both cells of a face take the SAME flux value).
    // (with the RK average the waiting value is already the weighted sum base (1 - w) + (...) w, and the upper-face term comes in with cy w:
    // five doubles wait, not ten)
    State5 pending;
    const double cy_last = COMBINE ? p.cy * p.weight : p.cy;
    auto complete_plane = [&] (int rr, int buf) __attribute__((always_inline))
    {
        const State5 Fy_hi = lds_get(tile.F[buf][row], lane);
        State5 Un;
#pragma unroll
        for (int q = 0; q < 5; ++q) Un[q] = __builtin_fma(-Fy_hi[q], cy_last, pending[q]);
        const bool bad_density = writes && !(Un[0] > 0.0);
        if (__any(bad_density))
        {
            if (bad_density) acc.note_value(Un[0], MH_STATUS_NEG_DENSITY, (uint32_t) rr * planeu + cellu);
        }
        if (writes)
        {
            store_plane(p.u_out + row_off(rr), plane, cw, Un);
            if (rr < H3 || rr >= p.n0 - H3)          // keep the physical axis-0 ghost planes of the output current (wave-uniform, cold)
            {
                if (p.bc_lo0 == 0 && rr == 0) { store_plane(p.u_out + row_off(-1), plane, cw, Un); store_plane(p.u_out + row_off(-2), plane, cw, Un); }
                if (p.bc_hi0 == 1 && rr < H3) store_plane(p.u_out + row_off(p.n0 + rr), plane, cw, Un);
                if (p.bc_hi0 == 0 && rr == p.n0 - 1) { store_plane(p.u_out + row_off(p.n0), plane, cw, Un); store_plane(p.u_out + row_off(p.n0 + 1), plane, cw, Un); }
                if (p.bc_lo0 == 1 && rr >= p.n0 - H3) store_plane(p.u_out + row_off(rr - p.n0), plane, cw, Un);
            }
        }
    };

    // One plane. ONE workgroup barrier: before it every wave publishes the primitives of its row of plane r (and the helper
    // waves those of the outside rows) and does the axis-0 and axis-2 work, which needs no other wave; after it every wave
    // reads its four axis-1 neighbours and computes BOTH of its axis-1 face fluxes itself. The face between two tile rows
    // is thereby evaluated twice (by identical code on identical inputs, so conservation holds bit for bit): that costs a
    // fourth Riemann problem per cell, but removes the two further exchanges (face states, fluxes) with their barriers and
    // the imbalance of an edge wave computing the tile's ninth face.
    auto plane_step = [&] (int r, auto k0) __attribute__((always_inline))
    {
        constexpr int K0 = decltype(k0)::value, K1 = (K0 + 1) % 3, K2 = (K0 + 2) % 3;
        const int pb = r & 1;

        const State5 Unext = load_plane(in + row_off(min(r + 3, p.n0 + 1)), plane, c0);
        State5 Ue;
        if (helper) Ue = load_plane(in + row_off(r), plane, ce);

        lds_put(tile.P[pb][row + H3], lane, P[K0]);

        // ---- axis 0: flux through face r+1/2
        P[K2] = A::c2p(Uin, gl);
        const bool bad_pressure = writes && !(P[K2][4] >= 0.0);
        lds_put(tile.U[row][K2], lane, Uin);
        Uin = Unext;
        if constexpr (PLM)
        {
            G[K1] = A::plm(P[K0], P[K1], P[K2], lim);
            Fx[K1] = A::template flux<RIEMANN, 0>(A::plus(P[K0], G[K0], lim), A::minus(P[K1], G[K1], lim), gl);
        }
        else
        {
            Fx[K1] = A::template flux<RIEMANN, 0>(P[K0], P[K1], gl);
        }

        // ---- axis 2 (lanes): this lane computes the flux through its LEFT face
        State5 Fz_lo, Fz_hi;
        if constexpr (PLM && A::shared_differences)
        {
            // the limiter's one-sided differences belong to a face: formed once, the left one comes from the left neighbour's lane
            const State5 Dr = A::difference(P[K0], dpp3_right(P[K0]));
            const State5 Gz = A::plm_from_differences(dpp3_left(Dr), Dr, lim);
            const State5 SL = dpp3_left(A::plus(P[K0], Gz, lim));
            Fz_lo = A::template flux<RIEMANN, 2>(SL, A::minus(P[K0], Gz, lim), gl);
        }
        else if constexpr (PLM)
        {
            const State5 Gz = A::plm(dpp3_left(P[K0]), P[K0], dpp3_right(P[K0]), lim);
            const State5 SL = dpp3_left(A::plus(P[K0], Gz, lim));
            Fz_lo = A::template flux<RIEMANN, 2>(SL, A::minus(P[K0], Gz, lim), gl);
        }
        else
        {
            Fz_lo = A::template flux<RIEMANN, 2>(dpp3_left(P[K0]), P[K0], gl);
        }
        Fz_hi = dpp3_right(Fz_lo);

        // deferred axis-1 mode: the update with the axis-0 and axis-2 terms already in it crosses the barrier (5 doubles instead of 4 fluxes)
        State5 Uxz;
        if constexpr (DEFER)
        {
            const State5 Ucur = lds_get(tile.U[row][K0], lane);
#pragma unroll
            for (int q = 0; q < 5; ++q) Uxz[q] = __builtin_fma(-(Fz_hi[q] - Fz_lo[q]), p.cz, __builtin_fma(-(Fx[K1][q] - Fx[K0][q]), p.cx, Ucur[q]));
        }

        if (helper) lds_put(tile.P[pb][eslot], lane, A::c2p(Ue, gl));
#ifndef MH_E3D_NOBARRIER      // diagnostic builds only: wrong results, shows what the barrier costs
        __syncthreads();
#endif
        if constexpr (DEFER)
        {
            // ---- the plane that has been waiting for its upper-face flux
            if (r > r0) complete_plane(r - 1, pb ^ 1);
            State5 Ubase;
            if constexpr (COMBINE) Ubase = load_plane(p.u_base + row_off(r), plane, c0);
            // ---- axis 1: the lower face of this row (the top wave: the tile's top face too), handed to the row below through LDS
            State5 Fy_lo;
            const State5 Pm1 = lds_get(tile.P[pb][row + H3 - 1], lane);
            if constexpr (PLM)
            {
                const State5 Pp1 = lds_get(tile.P[pb][row + H3 + 1], lane);
                const State5 Dm = A::difference(Pm1, P[K0]), Dp = A::difference(P[K0], Pp1);
                const State5 Gy = A::plm_from_differences(Dm, Dp, lim);
                {
                    const State5 Pm2 = lds_get(tile.P[pb][row + H3 - 2], lane);
                    Fy_lo = A::template flux<RIEMANN, 1>(A::plus(Pm1, A::plm_from_differences(A::difference(Pm2, Pm1), Dm, lim), lim), A::minus(P[K0], Gy, lim), gl);
                }
                if (row == ROWS3 - 1)
                {
                    const State5 Pp2 = lds_get(tile.P[pb][row + H3 + 2], lane);
                    lds_put(tile.F[pb][ROWS3 - 1], lane,
                            A::template flux<RIEMANN, 1>(A::plus(P[K0], Gy, lim), A::minus(Pp1, A::plm_from_differences(Dp, A::difference(Pp1, Pp2), lim), lim), gl));
                }
            }
            else
            {
                Fy_lo = A::template flux<RIEMANN, 1>(Pm1, P[K0], gl);
                if (row == ROWS3 - 1) lds_put(tile.F[pb][ROWS3 - 1], lane, A::template flux<RIEMANN, 1>(P[K0], lds_get(tile.P[pb][row + H3 + 1], lane), gl));
            }
            if (row > 0) lds_put(tile.F[pb][row - 1], lane, Fy_lo);

            // ---- this plane's update up to the upper-face term
#pragma unroll
            for (int q = 0; q < 5; ++q)
            {
                const double u1 = __builtin_fma(Fy_lo[q], p.cy, Uxz[q]);
                if constexpr (COMBINE) pending[q] = A::combine(Ubase[q], u1, p.weight);
                else                   pending[q] = u1;
            }
            if (__any(bad_pressure))
            {
                if (bad_pressure && r + 2 < p.n0) acc.note_value(P[K2][4], MH_STATUS_NEG_PRESSURE, (uint32_t) (r + 2) * planeu + cellu);
            }
            return;
        }
        State5 Ubase;
        if constexpr (COMBINE) Ubase = load_plane(p.u_base + row_off(r), plane, c0);

        // ---- axis 1 (across waves): both faces of this row
        State5 Fy_lo, Fy_hi;
        {
            const State5 Pm1 = lds_get(tile.P[pb][row + H3 - 1], lane), Pp1 = lds_get(tile.P[pb][row + H3 + 1], lane);
            if constexpr (PLM && A::shared_differences)
            {
                // three slopes (this row's and its two neighbours') from four differences instead of six
                const State5 Dm = A::difference(Pm1, P[K0]), Dp = A::difference(P[K0], Pp1);
                const State5 Gy = A::plm_from_differences(Dm, Dp, lim);
                {
                    const State5 Pm2 = lds_get(tile.P[pb][row + H3 - 2], lane);
                    Fy_lo = A::template flux<RIEMANN, 1>(A::plus(Pm1, A::plm_from_differences(A::difference(Pm2, Pm1), Dm, lim), lim), A::minus(P[K0], Gy, lim), gl);
                }
                {
                    const State5 Pp2 = lds_get(tile.P[pb][row + H3 + 2], lane);
                    Fy_hi = A::template flux<RIEMANN, 1>(A::plus(P[K0], Gy, lim), A::minus(Pp1, A::plm_from_differences(Dp, A::difference(Pp1, Pp2), lim), lim), gl);
                }
            }
            else if constexpr (PLM)
            {
                const State5 Gy = A::plm(Pm1, P[K0], Pp1, lim);
                {
                    const State5 Pm2 = lds_get(tile.P[pb][row + H3 - 2], lane);
                    Fy_lo = A::template flux<RIEMANN, 1>(A::plus(Pm1, A::plm(Pm2, Pm1, P[K0], lim), lim), A::minus(P[K0], Gy, lim), gl);
                }
                {
                    const State5 Pp2 = lds_get(tile.P[pb][row + H3 + 2], lane);
                    Fy_hi = A::template flux<RIEMANN, 1>(A::plus(P[K0], Gy, lim), A::minus(Pp1, A::plm(P[K0], Pp1, Pp2, lim), lim), gl);
                }
            }
            else
            {
                Fy_lo = A::template flux<RIEMANN, 1>(Pm1, P[K0], gl);
                Fy_hi = A::template flux<RIEMANN, 1>(P[K0], Pp1, gl);
            }
        }

        // ---- update (+ RK combine)
        const State5 Ucur = lds_get(tile.U[row][K0], lane);
        State5 Un;
#pragma unroll
        for (int q = 0; q < 5; ++q)
        {
            const double u1 = A::update3(Ucur[q], Fx[K0][q], Fx[K1][q], Fy_lo[q], Fy_hi[q], Fz_lo[q], Fz_hi[q], p.cx, p.cy, p.cz);
            if constexpr (COMBINE) Un[q] = A::combine(Ubase[q], u1, p.weight);
            else                   Un[q] = u1;
        }
        const bool bad_density = writes && !(Un[0] > 0.0);
        if (__any(bad_pressure || bad_density))
        {
            if (writes && bad_pressure && r + 2 < p.n0) acc.note_value(P[K2][4], MH_STATUS_NEG_PRESSURE, (uint32_t) (r + 2) * planeu + cellu);
            if (writes && bad_density) acc.note_value(Un[0], MH_STATUS_NEG_DENSITY, (uint32_t) r * planeu + cellu);
        }

        if (writes)
        {
            store_plane(p.u_out + row_off(r), plane, cw, Un);
            if (r < H3 || r >= p.n0 - H3)          // keep the physical axis-0 ghost planes of the output current (wave-uniform, cold)
            {
                if (p.bc_lo0 == 0 && r == 0) { store_plane(p.u_out + row_off(-1), plane, cw, Un); store_plane(p.u_out + row_off(-2), plane, cw, Un); }
                if (p.bc_hi0 == 1 && r < H3) store_plane(p.u_out + row_off(p.n0 + r), plane, cw, Un);
                if (p.bc_hi0 == 0 && r == p.n0 - 1) { store_plane(p.u_out + row_off(p.n0), plane, cw, Un); store_plane(p.u_out + row_off(p.n0 + 1), plane, cw, Un); }
                if (p.bc_lo0 == 1 && r >= p.n0 - H3) store_plane(p.u_out + row_off(r - p.n0), plane, cw, Un);
            }
        }
    };

    int r = r0;
    for (; r + 3 <= r1; r += 3)
    {
        plane_step(r, std::integral_constant<int, 0>());
        plane_step(r + 1, std::integral_constant<int, 1>());
        plane_step(r + 2, std::integral_constant<int, 2>());
    }
    if (r < r1) plane_step(r, std::integral_constant<int, 0>());
    if (r + 1 < r1) plane_step(r + 1, std::integral_constant<int, 1>());
    if constexpr (DEFER)
    {
        // the chunk's last plane: one more barrier makes its upper-face fluxes visible
        __syncthreads();
        complete_plane(r1 - 1, (r1 - 1) & 1);
    }

    acc.commit(p.status);
}

template<class A, int RIEMANN, bool PLM, bool COMBINE>
inline hipError_t launch3(const Stage3dParams& p, int nblocks, hipStream_t stream)
{
    auto kernel = euler3d_stage_kernel<A, RIEMANN, PLM, COMBINE>;
    // 120 KB of LDS per workgroup (deferred axis-1 mode: all 160 KB): dynamic + opt-in, once per DEVICE (a process may hold contexts on several)
    static std::atomic<uint64_t> attr_set_on(0);
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (! (attr_set_on.load(std::memory_order_acquire) & bit))
    {
        hipError_t e = hipFuncSetAttribute((const void*) kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) tile3d_bytes<A>());
        if (e != hipSuccess) return e;
        attr_set_on.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(kernel, dim3(nblocks), dim3(W3 * A::tile_rows), tile3d_bytes<A>(), stream, p);
    return hipGetLastError();
}

// the eight FAST instantiations (key = HLLC 4 | PLM 2 | COMBINE 1), compiled in euler3d_fast.hip
hipError_t euler3d_launch_fast(int key, const Stage3dParams& p, int nblocks, hipStream_t stream);

} // namespace mh
