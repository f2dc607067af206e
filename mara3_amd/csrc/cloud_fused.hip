// Both stages of an RK2 step of the `cloud` sub-program (2-D axisymmetric spherical-polar SRHD, BASELINE config 4) in ONE launch
// (gfx950 / MI355X), MH_ARITH_FAST + PLM: the whole field (nozzle inflow inside, zero gradient outside) and, since round 5, a radial SLAB of
// it whose cut sides are MH_BC_EXTERNAL (four stored rows of the neighbour per cut side, one exchange per step: slab.hip).
//
// What it replaces: `s0 * 0.5 + advance(advance(s0)) * 0.5` (src/subprog_cloud.cpp:676-697, `advance` :511-584) as two launches of
// cloud_stage_kernel (cloud.hip). Both stages use the nozzle row of the step-START time (:466-493, :524) and the same geometry, so the
// second stage needs nothing from the host between the two: the first-stage field lives in LDS only.
//
// Structure (that of euler2d_fused.hip): a PAIR of waves owns 64 polar columns and marches radially;
//   * the PRODUCER runs cloud.hip's first-stage row loop on the step-start field and leaves each row of u1 in a five-slot LDS ring;
//   * the CONSUMER runs the second-stage row loop on the rows of that ring (its loads are LDS reads), reads the step-start row for the RK
//     average from memory (as the second launch does; that read is not what bounds this kernel) and stores the result;
//   * one s_barrier per row keeps them in lockstep (LDS-only release: global loads and stores stay in flight across it);
//   * a WORKGROUP is two pairs on strips 60 columns apart: a consumer takes the two outermost first-stage columns per side, which its own
//     producer cannot form, from the neighbouring pair's ring - 116 output columns per workgroup.
// Physical radial sides cost no first-stage ghost rows: the inner ghost rows of BOTH stages are the nozzle primitives, the outer ones a copy of
// the last row's primitives, so the producer covers rows max(r0 - 2, 0) .. min(r1 + 2, nr) - 1 of a chunk [r0, r1) and nothing beyond the grid.
// Beyond a CUT the producer forms the neighbour's first-stage rows (two per side) from its four step-start rows, with the neighbour's geometry
// (row factors are indexed by GLOBAL row), so that every cell sees the bits of the one-domain run; a neighbour's cell that fails
// recover_primitive is the neighbour's to report.
// Every lane of the consumer converts a valid cell (ghost and out-of-range columns read the clamped column, lanes without a source a neighbour's):
// recover_primitive iterates, and one lane on garbage would hold its wave for fifty Newton steps.
//
// The arithmetic is SrhdFast's on the same values in the same order as the two launches: the result is bit-identical to theirs
// (tests/test_gpu_cloud_fused.py) and inherits their tolerance against the reference.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <type_traits>
#include "launch.hpp"
#include "status_device.hpp"
#include "euler_device.hpp"
#include "srhd_device.hpp"
#include "srhd_device_fast.hpp"
#include "cloud_rows.hpp"
#include "row_check.hpp"

namespace mh {

static constexpr int QWAVE = 64;
static constexpr int QHALO = 4;                      // two per stage
#ifndef MH_CLOUD_FUSED_PAIRS
#define MH_CLOUD_FUSED_PAIRS 2
#endif
static constexpr int QPAIRS = MH_CLOUD_FUSED_PAIRS;
static constexpr int QPITCH = QWAVE - 4;                                       // columns between neighbouring pairs of a workgroup
static constexpr int QGROUP = QPITCH * QPAIRS - 4;                             // output columns per workgroup: 56, 116
static constexpr int QSLOTS = 5;                     // hand-off ring: the consumer reads rows r .. r+2 (r0-2 .. r0+1 in its prologue) while the producer,
                                                     // at most one barrier ahead, writes row r+3
static constexpr int OSLOTS = 3;                     // the producer's own rows r .. r+2 waiting for their update (cloud.hip's ring)
// waves per SIMD the launch asks for: 3 = cloud.hip's (168 registers); 2 leaves the compiler 256
#ifndef MH_CLOUD_FUSED_WAVES
#define MH_CLOUD_FUSED_WAVES 3
#endif
// 1: the register window as three-slot rings with the row loop unrolled by three (no rotation moves; wants the 256 registers of two waves per
// SIMD); 0: cloud.hip's rotating window (fits the 168 registers of three waves per SIMD)
#ifndef MH_CLOUD_FUSED_RINGS
#define MH_CLOUD_FUSED_RINGS (MH_CLOUD_FUSED_WAVES < 3)
#endif

struct CloudFusedParams
{
    const double* u_in;
    double*       u_out;
    const double* cotq;        // [nq]   tan(pi/2 - theta_c)
    const double* rowf;        // [nr_global][8] per-row factors (mh_cloud_pack_geometry)
    const double* colf;        // [nq][8]        per-column factors
    const double* inflow;      // [5][nq] primitives of the inner ghost rows, step-start time
    int32_t*      status;
    long   plane_stride, row_stride;
    int    n0, n1, row_offset;
    int    chunk_rows, nstrips, nchunks;
    // the rows this launch covers: chunks [0, seg0_chunks) cut rows [seg0_begin, seg0_end), the others [seg1_begin, seg1_end) - a radial slab with
    // neighbours runs both of its edge strips in one launch and the rest in another (slab.hip); a whole field is one segment
    int    seg0_begin, seg0_end, seg0_chunks, seg1_begin, seg1_end;
    int    seg1_chunk_rows;       // rows per chunk of the second segment (= chunk_rows unless the launcher tapers the launch, see the launcher)
    int    ext_lo, ext_hi;        // 1: that radial side is a CUT of a slab decomposition (MH_BC_EXTERNAL) - rows -4 .. -1 / n0 .. n0 + 3 of u_in hold the
                                  // neighbour's step-start rows (four per side: two per stage), the first-stage rows beyond the cut are recomputed here
                                  // from them, and neither the nozzle rows nor the zero-gradient copy nor the zeroed edge slope apply on that side
    double gamma, theta, tfloor, dt;
};

__device__ inline void cloud_pair_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// PLANAR: field and nozzle row carry no azimuthal momentum (the `cloud` problem as upstream sets it up; mh_cloud_desc.planar, verified by the
// stepper at upload / mh_cloud_set_inflow): that component is neither read nor exchanged nor computed, and is written as zero - the other four
// components keep their bits, ~9 % fewer VALU instructions
template<bool PLANAR>
__global__ __launch_bounds__(2 * QWAVE * QPAIRS, MH_CLOUD_FUSED_WAVES)
void cloud_fused_rk2_kernel(CloudFusedParams p)
{
    using S = SrhdFastT<PLANAR>;
    constexpr auto live = [] (int q) { return ! (PLANAR && q == 3); };
    constexpr int NV = PLANAR ? 4 : 5;                            // variables held in the rings
    constexpr auto vi = [] (int q) { return PLANAR && q == 4 ? 3 : q; };
    __shared__ double hand_all[QPAIRS][QSLOTS][NV][QWAVE];       // first-stage rows on their way from the producer to the consumer
    __shared__ double own_all[QPAIRS][OSLOTS][NV][QWAVE];        // step-start rows waiting for the producer's update

    // the blocks of the first segment come first in launch order (what is launched last starts last); the XCD-aware order applies within a segment
    const int seg0_blocks = p.seg0_chunks * p.nstrips;
    const bool second = (int) blockIdx.x >= seg0_blocks;
    int b = second ? (int) blockIdx.x - seg0_blocks : (int) blockIdx.x;
    {
        const int per_xcd = (second ? (int) gridDim.x - seg0_blocks : seg0_blocks) >> 3;
        if (b < per_xcd * 8) b = (b & 7) * per_xcd + (b >> 3);      // neighbouring strips and chunks on one XCD
    }
    const int group = __builtin_amdgcn_readfirstlane(b);
    const int wave_of_group = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
#ifdef MH_CLOUD_FUSED_STAGGER      // probe: the workgroups that share a CU (ids about 256 apart) start a fraction of a row period apart
    for (int i = 0; i < (int) ((blockIdx.x >> 8) % 3) * MH_CLOUD_FUSED_STAGGER; ++i) __builtin_amdgcn_s_sleep(32);
#endif
    const int role = wave_of_group & 1;
    const int pp = wave_of_group >> 1;                  // which pair of the workgroup
    const int lane = threadIdx.x & 63;
    const int chunk = group / p.nstrips;                // within its segment
    const int strip = group - chunk * p.nstrips;
    const int n0 = p.n0, n1 = p.n1;
    const int chunk_rows = second ? p.seg1_chunk_rows : p.chunk_rows;
    const int r0 = (second ? p.seg1_begin : p.seg0_begin) + chunk * chunk_rows;
    const int r1 = min(r0 + chunk_rows, second ? p.seg1_end : p.seg0_end);
    const bool lo_phys = p.ext_lo == 0, hi_phys = p.ext_hi == 0;          // (wave-uniform)
    // rows of the first-stage field this chunk needs: nothing beyond the grid on a physical side, two rows of the neighbour's beyond a cut
    const int a_begin = lo_phys ? max(r0 - 2, 0) : r0 - 2, a_end = hi_phys ? min(r1 + 2, n0) : r1 + 2;
    const int row_min = lo_phys ? 0 : -4, row_max = hi_phys ? n0 - 1 : n0 + 3;      // step-start rows that exist

    const int col = strip * QGROUP - QHALO + pp * QPITCH + lane;
    const int jc = min(max(col, 0), n1 - 1);
    const bool pole_lo = col == 0, pole_hi = col == n1 - 1;

    const srhd::Gamma g = srhd::make_gamma(p.gamma);
    const double tfloor = p.tfloor;
    const typename S::Limiter lim = S::limiter(p.theta);
    const double cot = p.cotq[jc];
    double col_ar, col_dv, col_inv_dv, col_aq_lo, col_aq_hi;
    {
        const double* cf = p.colf + 8L * jc;
        col_ar = cf[0]; col_dv = cf[1]; col_inv_dv = cf[2]; col_aq_lo = cf[3]; col_aq_hi = cf[4];
    }
    // the poles (extend_zeros on the polar slopes and fluxes, :563 / :570): zero lane constants, as cloud.hip's FAST kernel
    typename S::Limiter lim_polar = lim;
    if (pole_lo) col_aq_lo = 0.0;
    if (pole_hi) col_aq_hi = 0.0;
    if (pole_lo || pole_hi) lim_polar.half_theta = 0.0;
    auto cell_geometry = [&] (int i) -> CellGeom
    {
        CellGeom c;
        const double* rf = p.rowf + 8L * i;          // wave-uniform: scalar loads
        c.dv = rf[2] * col_dv;
        c.inv_dv = rf[3] * col_inv_dv;
        c.nAr_lo = -(rf[0] * col_ar);
        c.nAr_hi = -(rf[1] * col_ar);
        c.nAq_lo = -(rf[4] * col_aq_lo);
        c.nAq_hi = -(rf[4] * col_aq_hi);
        c.rc = rf[5];
        c.inv_rc = rf[6];
        return c;
    };

    const long row_stride = p.row_stride, plane = p.plane_stride;
    auto row_off = [row_stride, n0, row_min, row_max] (int r) { (void) n0; (void) row_min; (void) row_max; return (long) (MH_ROW(r, min(row_min, -2), max(row_max, n0 + 1)) + 2) * row_stride; };
    const unsigned jc8 = (unsigned) jc * 8u;
    const uint32_t n1u = (uint32_t) n1, colu = (uint32_t) col;
    StatusAcc acc;

    auto inflow_row = [&] () -> State5
    {
        State5 P;
#pragma unroll
        for (int q = 0; q < 5; ++q) P[q] = live(q) ? p.inflow[(long) q * n1 + jc] : 0.0;
        return P;
    };
    // the five variables of a stored row (planar: four)
    auto load5 = [&] (const double* row) -> State5
    {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(row), 0, (int) (5 * plane * 8), 0x00020000);
        State5 U;
#pragma unroll
        for (int q = 0; q < 5; ++q)
            U[q] = live(q) ? __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, jc8, (unsigned) (q * plane * 8), 0)) : 0.0;
        return U;
    };
    auto lane_left = [&] (const State5& s) -> State5 { State5 r; for (int q = 0; q < 5; ++q) r[q] = live(q) ? dpp_left(s[q]) : 0.0; return r; };
    auto lane_right = [&] (const State5& s) -> State5 { State5 r; for (int q = 0; q < 5; ++q) r[q] = live(q) ? dpp_right(s[q]) : 0.0; return r; };

    if (role == 0)
    {
        // ================================================================ PRODUCER: first stage, rows a_begin .. a_end - 1 ================
        // (cloud.hip's row loop with COMBINE = false; the row goes to the hand-off ring instead of memory)
        double (*hand)[NV][QWAVE] = hand_all[pp];
        double (*own)[NV][QWAVE] = own_all[pp];
        const double* in = p.u_in;
        const bool real_col = lane >= 2 && lane < QWAVE - 2 && col >= 0 && col < n1;       // a cell of the grid whose first-stage value is valid here
        // the row loop requests rows up to three beyond the one it updates: held to the rows that exist (physical side: the rows of the grid -
        // the stored ghost rows hold nothing there; cut: the four rows of the neighbour)
        auto load_raw = [&] (int r) -> State5
        {
            const int rr = min(max(r, row_min), row_max);
            return load5(in + row_off(rr));
        };
        auto ring_put = [&] (int slot, const State5& raw)
        {
#pragma unroll
            for (int q = 0; q < 5; ++q) if (live(q)) own[slot][vi(q)][lane] = raw[q];
        };
        auto ring_get = [&] (int slot) -> State5
        {
            State5 U;
#pragma unroll
            for (int q = 0; q < 5; ++q) U[q] = live(q) ? own[slot][vi(q)][lane] : 0.0;
            return U;
        };
        auto prim_of_raw = [&] (int r, const State5& raw) -> State5
        {
            const CellGeom c = cell_geometry(p.row_offset + r);
            double x[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) x[q] = raw[q];
            S::to_density(x, c.dv, c.inv_dv);
            State5 U, P;
#pragma unroll
            for (int q = 0; q < 5; ++q) U[q] = x[q];
            const int st = S::c2p(U, g, tfloor, P);
            if (__any(st != 0))
            {
                if (real_col && st != 0 && r >= 0 && r < n0) acc.note((uint32_t) st, (uint32_t) r * n1u + colu);      // (a neighbour's row is the neighbour's to report)
            }
            return P;
        };
        auto prim_bc_raw = [&] (int r, const State5& last, const State5& raw) -> State5
        {
            if (lo_phys && r < 0) return inflow_row();
            if (hi_phys && r >= n0) return last;         // zero-gradient outer: copy of the last real row
            return prim_of_raw(r, raw);
        };
        auto prim_bc = [&] (int r, const State5& last) -> State5
        {
            if (lo_phys && r < 0) return inflow_row();
            if (hi_phys && r >= n0) return last;
            return prim_of_raw(r, load_raw(r));
        };

#if MH_CLOUD_FUSED_RINGS
        // register window as three-slot rings (slot = (row - a_begin) mod 3, compile-time in the loop unrolled by three): no
        // register-to-register rotation of the primitives, slopes, radial fluxes and the row in flight
        State5 U[3], P[3], G[3], Fx[3];
        {
            State5 dummy = {};
            const State5 Pb = prim_bc(a_begin - 1, dummy);
            const State5 raw0 = load_raw(a_begin);
            U[1] = load_raw(a_begin + 1);
            ring_put(0, raw0);
            ring_put(1, U[1]);
            P[0] = prim_of_raw(a_begin, raw0);
            P[1] = prim_bc_raw(a_begin + 1, P[0], U[1]);
            G[0] = S::plm(Pb, P[0], P[1], lim);
            State5 Gb;
            if (lo_phys && a_begin == 0) Gb = times_zero(G[0]);              // extend_zeros on G
            else              Gb = S::plm(prim_bc(a_begin - 2, dummy), Pb, P[0], lim);
            Fx[0] = S::template hlle<0>(S::plus(Pb, Gb, lim), S::minus(P[0], G[0], lim), g);
        }
        U[2] = load_raw(a_begin + 2);
        int hslot = (a_begin - (r0 - 2)) % QSLOTS;      // hand-off slot of row r: (r - (r0 - 2)) mod QSLOTS

        auto row_step = [&] (int r, auto k0) __attribute__((always_inline))
        {
            constexpr int K0 = decltype(k0)::value, K1 = (K0 + 1) % 3, K2 = (K0 + 2) % 3;
            U[K0] = load_raw(r + 3);                     // (row r's own values wait in the LDS ring)

            P[K2] = prim_bc_raw(r + 2, P[K1], U[K2]);
            ring_put(K2, U[K2]);
            if (hi_phys && r + 1 == n0) G[K1] = times_zero(G[K0]);
            else             G[K1] = S::plm(P[K0], P[K1], P[K2], lim);
            Fx[K1] = S::template hlle<0>(S::plus(P[K0], G[K0], lim), S::minus(P[K1], G[K1], lim), g);

            const State5 Gy = S::plm(lane_left(P[K0]), P[K0], lane_right(P[K0]), lim);          // (a pole lane's lim_polar gives it no weight)
            const State5 SL = lane_left(S::plus(P[K0], Gy, lim_polar));
            const State5 Fy_lo = S::template hlle<1>(SL, S::minus(P[K0], Gy, lim_polar), g);
            const State5 Fy_hi = lane_right(Fy_lo);

            const State5 U0 = ring_get(K0);
            __builtin_amdgcn_sched_barrier(0);
            const CellGeom c = cell_geometry(p.row_offset + r);
            const State5 Src = S::source(P[K0], c.rc, c.inv_rc, cot, g);
#pragma unroll
            for (int q = 0; q < 5; ++q)
                if (live(q)) hand[hslot][vi(q)][lane] = S::update(U0[q], Fx[K0][q], Fx[K1][q], Fy_lo[q], Fy_hi[q], c.nAr_lo, c.nAr_hi, c.nAq_lo, c.nAq_hi, Src[q], c.dv, p.dt);
            cloud_pair_barrier();                            // row r is in the ring
            hslot = hslot == QSLOTS - 1 ? 0 : hslot + 1;
        };
        int r = a_begin;
        for (; r + 3 <= a_end; r += 3)
        {
            row_step(r, std::integral_constant<int, 0>());
            row_step(r + 1, std::integral_constant<int, 1>());
            row_step(r + 2, std::integral_constant<int, 2>());
        }
        if (r < a_end) row_step(r, std::integral_constant<int, 0>());
        if (r + 1 < a_end) row_step(r + 1, std::integral_constant<int, 1>());
#else
        State5 P0, P1, G0, Fx_lo;
        {
            State5 dummy = {};
            const State5 Pb = prim_bc(a_begin - 1, dummy);
            const State5 raw0 = load_raw(a_begin), raw1 = load_raw(a_begin + 1);
            ring_put(0, raw0);
            ring_put(1, raw1);
            P0 = prim_of_raw(a_begin, raw0);
            P1 = prim_bc_raw(a_begin + 1, P0, raw1);
            G0 = S::plm(Pb, P0, P1, lim);
            State5 Gb;
            if (lo_phys && a_begin == 0) Gb = times_zero(G0);                // extend_zeros on G
            else              Gb = S::plm(prim_bc(a_begin - 2, dummy), Pb, P0, lim);
            Fx_lo = S::template hlle<0>(S::plus(Pb, Gb, lim), S::minus(P0, G0, lim), g);
        }

        State5 Uahead = load_raw(a_begin + 2);
        int slot = 0;
        int hslot = (a_begin - (r0 - 2)) % QSLOTS;      // hand-off slot of row r: (r - (r0 - 2)) mod QSLOTS
        for (int r = a_begin; r < a_end; ++r)
        {
            const State5 Unext = load_raw(r + 3);

            const State5 P2 = prim_bc_raw(r + 2, P1, Uahead);
            ring_put(slot == 0 ? 2 : slot - 1, Uahead);
            Uahead = Unext;
            State5 G1;
            if (hi_phys && r + 1 == n0) G1 = times_zero(G0);
            else             G1 = S::plm(P0, P1, P2, lim);
            const State5 Fx_hi = S::template hlle<0>(S::plus(P0, G0, lim), S::minus(P1, G1, lim), g);

            const State5 Gy = S::plm(lane_left(P0), P0, lane_right(P0), lim);          // (a pole lane's lim_polar gives it no weight)
            const State5 SL = lane_left(S::plus(P0, Gy, lim_polar));
            const State5 Fy_lo = S::template hlle<1>(SL, S::minus(P0, Gy, lim_polar), g);
            const State5 Fy_hi = lane_right(Fy_lo);

            const State5 U0 = ring_get(slot);
            __builtin_amdgcn_sched_barrier(0);
            const CellGeom c = cell_geometry(p.row_offset + r);
            const State5 Src = S::source(P0, c.rc, c.inv_rc, cot, g);
#pragma unroll
            for (int q = 0; q < 5; ++q)
                if (live(q)) hand[hslot][vi(q)][lane] = S::update(U0[q], Fx_lo[q], Fx_hi[q], Fy_lo[q], Fy_hi[q], c.nAr_lo, c.nAr_hi, c.nAq_lo, c.nAq_hi, Src[q], c.dv, p.dt);
            cloud_pair_barrier();                            // row r is in the ring

            P0 = P1; P1 = P2;
            G0 = G1;
            Fx_lo = Fx_hi;
            slot = slot == 2 ? 0 : slot + 1;
            hslot = hslot == QSLOTS - 1 ? 0 : hslot + 1;
        }
#endif
    }
    else
    {
        // ================================================================ CONSUMER: second stage + RK average, rows r0 .. r1 - 1 ========
        // (cloud.hip's row loop with COMBINE = true, weight 1/2; its loads of the first-stage field are reads of the ring)
        const int out_lo = pp > 0 ? 2 : QHALO, out_hi = pp < QPAIRS - 1 ? QWAVE - 2 : QWAVE - QHALO;
        const bool writes = lane >= out_lo && lane < out_hi && col < n1;
        const unsigned col8 = (unsigned) (writes ? col : 0) * 8u;
        // the lane of the pair's coordinate system that holds this lane's column (ghost columns: the clamped one, as cloud.hip's loads)
        const int want = lane + (jc - col);
        int other = pp, src = want;
        if (want < 2 && pp > 0) { other = pp - 1; src = want + QPITCH; }
        else if (want > QWAVE - 3 && pp < QPAIRS - 1) { other = pp + 1; src = want - QPITCH; }
        if (src < 2 || src > QWAVE - 3) { other = pp; src = min(max(want, 2), QWAVE - 3); }     // no source: a valid neighbour's column (result unused)
        const double* const hand_flat = &hand_all[0][0][0][0];
        const int hand_off = other * (QSLOTS * NV * QWAVE) + src;
        auto hand_row = [&] (int rr) -> State5
        {
            const int slot = (rr - (r0 - 2)) % QSLOTS;
            State5 U;
#pragma unroll
            for (int q = 0; q < 5; ++q) U[q] = live(q) ? hand_flat[hand_off + (slot * NV + vi(q)) * QWAVE] : 0.0;
            return U;
        };
        auto prim_of_raw = [&] (int r, const State5& raw) -> State5
        {
            const CellGeom c = cell_geometry(p.row_offset + r);
            double x[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) x[q] = raw[q];
            S::to_density(x, c.dv, c.inv_dv);
            State5 U, P;
#pragma unroll
            for (int q = 0; q < 5; ++q) U[q] = x[q];
            const int st = S::c2p(U, g, tfloor, P);
            if (__any(st != 0))
            {
                if (writes && st != 0 && r >= 0 && r < n0) acc.note((uint32_t) st, (uint32_t) r * n1u + colu);
            }
            return P;
        };
        auto prim_bc = [&] (int r, const State5& last) -> State5
        {
            if (lo_phys && r < 0) return inflow_row();
            if (hi_phys && r >= n0) return last;
            return prim_of_raw(r, hand_row(r));
        };

        const int fill = min(r0 + 2, a_end) - a_begin;          // barriers until rows a_begin .. min(r0 + 1, a_end - 1) are in the ring
        for (int k = 0; k < fill; ++k) cloud_pair_barrier();

#if MH_CLOUD_FUSED_RINGS
        State5 P[3], G[3], Fx[3];
        {
            State5 dummy = {};
            const State5 Pb = prim_bc(r0 - 1, dummy);
            P[0] = prim_of_raw(r0, hand_row(r0));
            P[1] = prim_bc(r0 + 1, P[0]);
            G[0] = S::plm(Pb, P[0], P[1], lim);
            State5 Gb;
            if (lo_phys && r0 == 0) Gb = times_zero(G[0]);
            else         Gb = S::plm(prim_bc(r0 - 2, dummy), Pb, P[0], lim);
            Fx[0] = S::template hlle<0>(S::plus(Pb, Gb, lim), S::minus(P[0], G[0], lim), g);
        }

        auto row_step = [&] (int r, auto k0) __attribute__((always_inline))
        {
            constexpr int K0 = decltype(k0)::value, K1 = (K0 + 1) % 3, K2 = (K0 + 2) % 3;
            if (r + 2 < a_end) cloud_pair_barrier();         // row r + 2 is in the ring
            P[K2] = prim_bc(r + 2, P[K1]);
            if (hi_phys && r + 1 == n0) G[K1] = times_zero(G[K0]);
            else             G[K1] = S::plm(P[K0], P[K1], P[K2], lim);
            Fx[K1] = S::template hlle<0>(S::plus(P[K0], G[K0], lim), S::minus(P[K1], G[K1], lim), g);

            const State5 Gy = S::plm(lane_left(P[K0]), P[K0], lane_right(P[K0]), lim);
            const State5 SL = lane_left(S::plus(P[K0], Gy, lim_polar));
            const State5 Fy_lo = S::template hlle<1>(SL, S::minus(P[K0], Gy, lim_polar), g);
            const State5 Fy_hi = lane_right(Fy_lo);

            const State5 U0 = hand_row(r);
            const State5 Ubase = load5(p.u_in + row_off(r));
            __builtin_amdgcn_sched_barrier(0);
            const CellGeom c = cell_geometry(p.row_offset + r);
            const State5 Src = S::source(P[K0], c.rc, c.inv_rc, cot, g);
            State5 Un;
#pragma unroll
            for (int q = 0; q < 5; ++q)
            {
                if (! live(q)) { Un[q] = 0.0; continue; }
                const double u1 = S::update(U0[q], Fx[K0][q], Fx[K1][q], Fy_lo[q], Fy_hi[q], c.nAr_lo, c.nAr_hi, c.nAq_lo, c.nAq_hi, Src[q], c.dv, p.dt);
                Un[q] = S::combine(Ubase[q], u1, 0.5);
            }
            if (writes) cloud_store_row(p.u_out + row_off(r), plane, col8, Un);
        };
        int r = r0;
        for (; r + 3 <= r1; r += 3)
        {
            row_step(r, std::integral_constant<int, 0>());
            row_step(r + 1, std::integral_constant<int, 1>());
            row_step(r + 2, std::integral_constant<int, 2>());
        }
        if (r < r1) row_step(r, std::integral_constant<int, 0>());
        if (r + 1 < r1) row_step(r + 1, std::integral_constant<int, 1>());
#else
        State5 P0, P1, G0, Fx_lo;
        {
            State5 dummy = {};
            const State5 Pb = prim_bc(r0 - 1, dummy);
            P0 = prim_of_raw(r0, hand_row(r0));
            P1 = prim_bc(r0 + 1, P0);
            G0 = S::plm(Pb, P0, P1, lim);
            State5 Gb;
            if (lo_phys && r0 == 0) Gb = times_zero(G0);
            else         Gb = S::plm(prim_bc(r0 - 2, dummy), Pb, P0, lim);
            Fx_lo = S::template hlle<0>(S::plus(Pb, Gb, lim), S::minus(P0, G0, lim), g);
        }

        for (int r = r0; r < r1; ++r)
        {
            if (r + 2 < a_end) cloud_pair_barrier();         // row r + 2 is in the ring
            const State5 P2 = prim_bc(r + 2, P1);
            State5 G1;
            if (hi_phys && r + 1 == n0) G1 = times_zero(G0);
            else             G1 = S::plm(P0, P1, P2, lim);
            const State5 Fx_hi = S::template hlle<0>(S::plus(P0, G0, lim), S::minus(P1, G1, lim), g);

            const State5 Gy = S::plm(lane_left(P0), P0, lane_right(P0), lim);
            const State5 SL = lane_left(S::plus(P0, Gy, lim_polar));
            const State5 Fy_lo = S::template hlle<1>(SL, S::minus(P0, Gy, lim_polar), g);
            const State5 Fy_hi = lane_right(Fy_lo);

            const State5 U0 = hand_row(r);
            const State5 Ubase = load5(p.u_in + row_off(r));
            __builtin_amdgcn_sched_barrier(0);
            const CellGeom c = cell_geometry(p.row_offset + r);
            const State5 Src = S::source(P0, c.rc, c.inv_rc, cot, g);
            State5 Un;
#pragma unroll
            for (int q = 0; q < 5; ++q)
            {
                if (! live(q)) { Un[q] = 0.0; continue; }
                const double u1 = S::update(U0[q], Fx_lo[q], Fx_hi[q], Fy_lo[q], Fy_hi[q], c.nAr_lo, c.nAr_hi, c.nAq_lo, c.nAq_hi, Src[q], c.dv, p.dt);
                Un[q] = S::combine(Ubase[q], u1, 0.5);
            }
            if (writes) cloud_store_row(p.u_out + row_off(r), plane, col8, Un);

            P0 = P1; P1 = P2;
            G0 = G1;
            Fx_lo = Fx_hi;
        }
#endif
    }
    acc.commit(p.status);
}

// how the last launch of this translation unit cut its rows: {chunk rows, chunk rows of the second segment, chunks of the first, chunks} (tests)
static int last_cut[4] = {0, 0, 0, 0};
void cloud_fused_last_cut(int out[4]) { for (int k = 0; k < 4; ++k) out[k] = last_cut[k]; }

// with_cuts: MH_BC_EXTERNAL radial sides are accepted too (a slab of a radial decomposition: the caller - slab.hip - keeps FOUR rows of the
// neighbour beyond such a side and exchanges once per step)
bool cloud_fused_rk2_available(const mh_cloud_desc* d, bool with_cuts)
{
    const bool lo_ok = d->bc_lo0 == MH_BC_INFLOW || (with_cuts && d->bc_lo0 == MH_BC_EXTERNAL);
    const bool hi_ok = d->bc_hi0 == MH_BC_OUTFLOW || (with_cuts && d->bc_hi0 == MH_BC_EXTERNAL);
    const bool whole = d->row_offset == 0 && d->nr == d->nr_global;
    return d->arith == MH_ARITH_FAST && d->plm_theta >= 0.0 && lo_ok && hi_ok && (with_cuts || whole) && d->nr >= 4 && d->nq >= 3
        && (d->bc_lo0 == MH_BC_INFLOW) == (d->row_offset == 0) && (d->bc_hi0 == MH_BC_OUTFLOW) == (d->row_offset + d->nr == d->nr_global);
}

// u_out = u_in * 0.5 + advance(advance(u_in)) * 0.5 over the whole field (layout of include/mara_hip.h; the two fields must differ);
// the nozzle row (inflow_dev, [5][nq] primitives) is that of the step-start time for both stages (src/subprog_cloud.cpp:466-493, :524)
hipError_t cloud_fused_rk2_launch(const mh_cloud_desc* d, const double* geom_dev, const double* inflow_dev, const double* u_in, double* u_out,
                                  double dt, int32_t* status, hipStream_t stream)
{
    return cloud_fused_rk2_launch_rows(d, geom_dev, inflow_dev, u_in, u_out, dt, 0, d->nr, 0, 0, status, stream, false, 0);
}

// ... over rows [a, b) and, in the same launch, [a2, b2) (b2 <= a2: none) of the field
hipError_t cloud_fused_rk2_launch_rows(const mh_cloud_desc* d, const double* geom_dev, const double* inflow_dev, const double* u_in, double* u_out,
                                       double dt, int a, int b, int a2, int b2, int32_t* status, hipStream_t stream, bool with_cuts, int late_blocks)
{
    if (! cloud_fused_rk2_available(d, with_cuts) || u_in == u_out || ! geom_dev || ! inflow_dev) return hipErrorInvalidValue;
    if (a < 0 || b > d->nr || b <= a || (b2 > a2 && (a2 < b || b2 > d->nr))) return hipErrorInvalidValue;
    int rows0 = b - a, rows1 = b2 > a2 ? b2 - a2 : 0;
    const int longest = rows0 > rows1 ? rows0 : rows1;
    CloudFusedParams p;
    p.u_in = u_in; p.u_out = u_out; p.status = status;
    p.n0 = d->nr; p.n1 = d->nq; p.row_offset = d->row_offset;
    // geom_dev: rv[nr_global+1] | dmu[nq] | sinq[nq+1] | cotq[nq] | rowf[nr_global][8] | colf[nq][8]
    p.cotq = geom_dev + (d->nr_global + 1) + d->nq + (d->nq + 1);
    p.rowf = p.cotq + d->nq;
    p.colf = p.rowf + 8L * d->nr_global;
    p.inflow = inflow_dev;
    p.plane_stride = d->nq;
    p.row_stride = 5L * d->nq;
    p.nstrips = (p.n1 + QGROUP - 1) / QGROUP;
    if (d->chunk_rows > 0) p.chunk_rows = d->chunk_rows;
    else
    {
        // as euler2d_fused.hip: a pair pays about eight pipeline-fill rows per chunk, and CUs x MH_CLOUD_FUSED_WAVES workgroups are resident
        // at a time: the shortest chunk that fills R residency rounds to the brim, for the smallest R that keeps it near 100 rows
        const int resident = device_cu_count() * (4 * MH_CLOUD_FUSED_WAVES) / (2 * QPAIRS);      // workgroups: 768 on 256 CUs
        int rounds = 1;
        auto chunk_for = [&] (int r) { const int nch = resident * r / p.nstrips > 0 ? resident * r / p.nstrips : 1; return (rows0 + rows1 + nch - 1) / nch; };
        while (chunk_for(rounds) > 80) ++rounds;      // measured at 4096^2 (gpurun_out/r4b, profiles/r04/ab_cloud_fused_chunks.jsonl): 64 rows (3.0 rounds) 0.993 ms, 98 (1.97) 1.010, 49 (3.9) 1.024, 196 (0.98) 1.037
        p.chunk_rows = chunk_for(rounds);
        if (p.chunk_rows < 8) p.chunk_rows = 8;
    }
    if (p.chunk_rows > longest) p.chunk_rows = longest;
    if (p.chunk_rows < 2 && longest >= 2) p.chunk_rows = 2;
    p.seg1_chunk_rows = p.chunk_rows;
    // TAPER, as euler2d_fused.hip's (where it is measured: the same 36 strips, 768 resident workgroups and 72 edge workgroups make the 4-GPU cut of
    // a 4096-row grid end 5 % sooner): the interior launch of a radial slab with neighbours - `late_blocks` slots are held by its edge launch when
    // it starts - ends in shorter chunks, launched last, so that the workgroups that start late do not end late. Config 4's four 1024-row slabs
    // are this case; on ONE GPU, where the four slabs share the device, it cannot be measured (profiles/r05/cloud_fused_cuts.md).
    // MH_FUSED_TAPER_ROWS as there (default 8 rows, 0 = off); from 24-row short chunks on.
    // (read per launch, not cached: tests/test_gpu_slab_group.py and test_gpu_cloud_fused.py exercise the tapered cut on small grids with
    // MH_FUSED_TAPER_MIN - the shortest short chunk for which the taper applies, default 24 rows)
    const int taper = [] { const char* v = getenv("MH_FUSED_TAPER_ROWS"); return v ? atoi(v) : 8; } ();
    const int taper_min = [] { const char* v = getenv("MH_FUSED_TAPER_MIN"); return v ? atoi(v) : 24; } ();
    if (late_blocks > 0 && taper > 0 && rows1 == 0 && d->chunk_rows <= 0)
    {
        const int resident = device_cu_count() * (4 * MH_CLOUD_FUSED_WAVES) / (2 * QPAIRS);
        const int nch = resident / p.nstrips;
        const int nshort = (late_blocks + p.nstrips - 1) / p.nstrips;
        const int clong = nch > 0 ? (rows0 + nshort * taper + nch - 1) / nch : 0;
        if (nch > nshort && clong - taper >= taper_min && clong <= 80)
        {
            const int long_rows = (nch - nshort) * clong;
            if (long_rows < rows0)
            {
                p.chunk_rows = clong;
                p.seg1_chunk_rows = clong - taper;
                a2 = a + long_rows; b2 = b;
                b = a2;
                rows0 = long_rows; rows1 = b2 - a2;
            }
        }
    }
    p.seg0_begin = a; p.seg0_end = b;
    p.seg0_chunks = (rows0 + p.chunk_rows - 1) / p.chunk_rows;
    p.seg1_begin = a2; p.seg1_end = rows1 ? b2 : a2;
    p.nchunks = p.seg0_chunks + (rows1 ? (rows1 + p.seg1_chunk_rows - 1) / p.seg1_chunk_rows : 0);
    last_cut[0] = p.chunk_rows; last_cut[1] = p.seg1_chunk_rows; last_cut[2] = p.seg0_chunks; last_cut[3] = p.nchunks;          // (mh_debug_last_fused_cut: tests)
    p.ext_lo = d->bc_lo0 == MH_BC_EXTERNAL ? 1 : 0;
    p.ext_hi = d->bc_hi0 == MH_BC_EXTERNAL ? 1 : 0;
    p.gamma = d->gamma; p.theta = d->plm_theta; p.tfloor = d->temperature_floor;
    p.dt = dt;
    const dim3 grid(p.nstrips * p.nchunks), block(2 * QWAVE * QPAIRS);
    // d->planar > 0: the caller (a stepper that has verified it) knows field and nozzle row to carry no azimuthal momentum
    if (d->planar > 0) hipLaunchKernelGGL(cloud_fused_rk2_kernel<true>, grid, block, 0, stream, p);
    else               hipLaunchKernelGGL(cloud_fused_rk2_kernel<false>, grid, block, 0, stream, p);
    return hipGetLastError();
}

int cloud_fused_rk2_blocks_per_chunk(const mh_cloud_desc* d) { return (d->nq + QGROUP - 1) / QGROUP; }

// row-range guard (row_check.hpp): what this translation unit's kernels asked for; false in product builds
MH_ROW_RANGE_READER(rows_requested_cloud_fused)

} // namespace mh
