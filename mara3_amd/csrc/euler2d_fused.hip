// Both stages of an RK2 step of the 2-D uniform-cartesian Euler scheme in ONE launch (gfx950 / MI355X), MH_ARITH_FAST + PLM only.
//
// What it replaces: `s0 * 0.5 + advance(advance(s0)) * 0.5` (src/subprog_cloud.cpp:676-697 with the `advance` of :511-584 specialised
// to mara::euler on a cartesian grid, oracle/ref_drivers/euler_cart_compose.hpp) as two launches of euler2d_stage_kernel, which move
// 80 + 120 B per zone-update through HBM (the first-stage field is written and read back, the step-start field is read twice). Here
// the first-stage field never exists in memory: 40 B read + 40 B written per zone-update.
//
// Structure: a PAIR of waves owns a strip of 64 columns and marches along axis 0:
//   * the PRODUCER wave runs the first stage - the row loop of euler2d.hip on the step-start field, two rows ahead - and, instead of
//     storing a row of u1, leaves it in a five-slot ring in LDS;
//   * the CONSUMER wave runs the second stage on the rows of that ring (its "loads" are LDS reads), reads the step-start row for the
//     RK average from a second LDS ring, where the producer parked it four rows earlier, and stores the result.
//   One s_barrier per row keeps them in lockstep; it waits for LDS only (the waves' global loads stay in flight across it). Both
//   waves keep the register footprint of a single-stage kernel, so the launch still holds two waves per SIMD, and the two kinds of wave -
//   one issue-bound, one with the stores - share every CU.
//   * A WORKGROUP is TWO such pairs on strips 60 columns apart (MH_FUSED_PAIRS). A producer's first-stage values are valid in its lanes
//     2 .. 61; a consumer needs them two lanes beyond its outputs, and where its own producer has none (lanes 0, 1, 62, 63) the
//     neighbouring pair's ring holds the column. So the two pairs overlap by the first stage's halo only: 58 output columns per pair
//     (116 per workgroup) where a lone pair has 56 - 4 % fewer pairs, and measured 5.5 % less time per 4096^2 step (0.582 against
//     0.616 ms on one box; four pairs per workgroup: 0.639, the barrier then holds eight waves and a CU one workgroup;
//     profiles/r03/ab_fused_pairs_per_workgroup.jsonl).
//   * Ghost cells of the FIRST-STAGE field need no pass over memory: a ghost COLUMN is the LDS read of another lane (outflow: the edge
//     lane; periodic: the lane already holds the wrapped column), a ghost ROW is another slot of the ring (outflow) or the producer's
//     own work on the wrapped row (periodic).
//   Redundancy against the two-launch form: 6 halo lanes of 64 instead of 4, and 4 + 4 pipeline-fill rows per chunk instead of 4.
//
// The arithmetic is FastArith's (euler_device_fast.hpp) on the same values in the same order as the two launches, so the result is
// bit-identical to theirs (tests/test_gpu_fused_rk2.py) and inherits their tolerance against the reference (L1 <= 1e-12).
// Not for: STRICT arithmetic (issue-bound at 2.5x the instructions: nothing to gain from less traffic), cut sides (MH_BC_EXTERNAL: the
// multi-rank steppers exchange the first-stage field), RK1, piecewise-constant reconstruction.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <type_traits>
#include "euler_device.hpp"
#include "euler_device_fast.hpp"
#include "launch.hpp"
#include "status_device.hpp"
#include "euler2d_rows.hpp"
#include "row_check.hpp"

namespace mh {

static constexpr int FWAVE = 64;
static constexpr int FHALO = 4;                      // two per stage
// MH_FUSED_PAIRS pairs make a workgroup, on neighbouring strips 60 columns apart. A consumer whose own producer has no valid first-stage
// value in its two outermost lanes per side takes them from the neighbouring pair's ring, so the pairs of a workgroup overlap by the
// FIRST stage's halo only: 60 output columns per inner pair and 58 per outer one instead of 56, one barrier for all waves per row.
#ifndef MH_FUSED_PAIRS
#define MH_FUSED_PAIRS 2
#endif
static constexpr int FPAIRS = MH_FUSED_PAIRS;
// MH_FUSED_MASK_HALO = 1: lanes whose result nobody uses sit out the axis-0 flux and the update (EXEC-masked: the instructions issue all the
// same, the lanes do not switch). Measured and NOT taken: 0.585 against 0.580 ms per 4096^2 step (profiles/r03/ab_fused_mask_halo.jsonl) -
// what the two-pair workgroup gained beyond its 4 % fewer pairs is not the idle lanes' switching.
#ifndef MH_FUSED_MASK_HALO
#define MH_FUSED_MASK_HALO 0
#endif
static constexpr int FPITCH = FWAVE - 4;                                       // columns between neighbouring pairs of a workgroup
static constexpr int FGROUP = FPITCH * FPAIRS - 4;                             // output columns per workgroup: 56, 116, 236
static constexpr int FSLOTS = 5;                     // hand-off ring. The consumer reads rows b .. b+2 (four rows in its prologue) while the producer, at most
                                                     // one barrier ahead, writes row b+3: five live slots
static constexpr int USLOTS = 6;                     // step-start rows b .. b+4 (the producer converts row b+4 while the consumer averages with row b), and one ahead

struct Fused2dParams
{
    const double* u_in;
    double*       u_out;
    int32_t*      status;
    long   plane_stride, row_stride;
    int    n0, n1;
    int    chunk_rows, nstrips, nchunks;
    // the rows this launch covers: chunks [0, seg0_chunks) cut rows [seg0_begin, seg0_end), the others [seg1_begin, seg1_end) - a slab with
    // neighbours runs both of its edge strips in one launch and the rest in another (slab.hip); a whole field is one segment
    int    seg0_begin, seg0_end, seg0_chunks, seg1_begin, seg1_end;
    int    seg1_chunk_rows;       // rows per chunk of the second segment (= chunk_rows unless the launcher tapers the launch: the interior of a slab
                                  // with neighbours ends in SHORTER chunks, whose workgroups start late - behind the edge launch's - and must not end late)
    int    bc0_lo, bc0_hi;        // axis 0, per side: 0 outflow, 1 periodic (both sides then), 2 EXTERNAL - a cut of a slab decomposition: rows
                                  // -4 .. -1 / n0 .. n0 + 3 of u_in hold the neighbour's rows (four per side: two per stage), nothing is
                                  // clamped or wrapped there, and the result's ghost rows on that side are the next exchange's to fill
    int    bc1;                   // axis 1: 0 outflow, 1 periodic
    double gamma, theta, cx, cy;
};

// Lane-to-lane movement of a row's five values: DPP moves (two VALU instructions per double) or, per exchange, the LDS crossbar
// (ds_bpermute_b32: no VALU issue, LDS latency). MH_FUSED_XCHG is a mask of the exchanges that go through the crossbar:
// 1 = primitives from the right, 2 = differences from the left, 4 = face states from the left, 8 = fluxes from the right.
#ifndef MH_FUSED_XCHG
#define MH_FUSED_XCHG 0
#endif
template<int BIT, bool PLANAR>
__device__ inline State5 lane_from(const State5& s, int addr_left, int addr_right, bool left)
{
    State5 r;
    if constexpr ((MH_FUSED_XCHG & BIT) != 0)
    {
        const int addr = left ? addr_left : addr_right;
#pragma unroll
        for (int q = 0; q < 5; ++q)
        {
            if (PLANAR && q == 3) { r[q] = 0.0; continue; }
            const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(s[q]));
            const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(s[q]));
            r[q] = __hiloint2double(hi, lo);
        }
    }
    else
    {
#pragma unroll
        for (int q = 0; q < 5; ++q)
        {
            if (PLANAR && q == 3) { r[q] = 0.0; continue; }
            r[q] = left ? from_left(s[q]) : from_right(s[q]);
        }
    }
    return r;
}

// LDS-only barrier of the pair: the waves' outstanding global loads and stores are not waited for
__device__ inline void pair_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
#ifndef MH_PROBE_NO_PAIR_BARRIER           // timing probe only (scripts/build_variant_one.sh): what the lockstep costs; the result is wrong without it
    __builtin_amdgcn_s_barrier();
#endif
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// PLANAR: the field's third momentum is identically zero (a 2-D run of the five-component state: the reference carries it as zeros) - the
// launcher's choice where the stepper has verified that at upload (mh_euler_cart_desc.planar). The component is neither read nor exchanged
// nor computed, and is written as zero: 72 instead of 80 B per cell and ~11 % fewer VALU instructions for the same bits in the other four.
// waves per SIMD the planar kernel is built for: its rings hold four variables (45 KB per workgroup) and it needs 158 - 170 registers, so a third
// wave per SIMD fits where the general kernel (186 - 208 registers, 56 KB) holds two
// (measured, profiles/r04/ab_planar.jsonl: at 158 registers and 45 KB the hardware places three workgroups per CU whatever the bound says;
// what the setting decides is the chunk length below - 64 rows = exactly three residency rounds of 768 workgroups: 0.471 ms per 4096^2 step,
// 98 rows 0.478, 48 / 80 / 128 rows 0.508 / 0.513 / 0.518)
#ifndef MH_FUSED_PLANAR_WAVES
#define MH_FUSED_PLANAR_WAVES 3
#endif
static constexpr int fused_waves_per_simd(bool planar) { return planar ? MH_FUSED_PLANAR_WAVES : 2; }

template<int RIEMANN, bool PLANAR>
__global__ __launch_bounds__(2 * FWAVE * FPAIRS, fused_waves_per_simd(PLANAR))
void euler2d_fused_rk2_kernel(Fused2dParams p)
{
    using A = FastArithT<PLANAR>;
    constexpr auto live = [] (int q) { return ! (PLANAR && q == 3); };
    constexpr int NV = PLANAR ? 4 : 5;                            // variables held in the rings
    constexpr auto vi = [] (int q) { return PLANAR && q == 4 ? 3 : q; };
    __shared__ double hand_all[FPAIRS][FSLOTS][NV][FWAVE];       // first-stage rows on their way from the producer to the consumer
    __shared__ double start_all[FPAIRS][USLOTS][NV][FWAVE];      // step-start rows: they wait for the producer's update (as euler2d.hip's ring) and for the consumer's average

    // the blocks of the first segment come first IN LAUNCH ORDER (the hardware starts workgroups in the order of their ids: what is launched last
    // starts last), the XCD-aware order applies within each segment
    const int seg0_blocks = p.seg0_chunks * p.nstrips;
    const bool second = (int) blockIdx.x >= seg0_blocks;
    int b = second ? (int) blockIdx.x - seg0_blocks : (int) blockIdx.x;
    {
        const int per_xcd = (second ? (int) gridDim.x - seg0_blocks : seg0_blocks) >> 3;
        if (b < per_xcd * 8) b = (b & 7) * per_xcd + (b >> 3);      // neighbouring strips and chunks on one XCD (halo re-reads hit its L2)
    }
    const int pair = __builtin_amdgcn_readfirstlane(b);
    const int wave_of_group = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
#ifdef MH_FUSED_ROLE_FLIP      // probe: which wave of a pair produces alternates between workgroups (bit MH_FUSED_ROLE_FLIP of the launch's block index)
    const int role = (wave_of_group ^ (((int) blockIdx.x >> MH_FUSED_ROLE_FLIP) & 1)) & 1;
#else
    const int role = wave_of_group & 1;
#endif
    const int pp = wave_of_group >> 1;                  // which pair of the workgroup
    double (*hand)[NV][FWAVE] = hand_all[pp];
    double (*start_rows)[NV][FWAVE] = start_all[pp];
    const int lane = threadIdx.x & 63;
    const int al = ((lane - 1) & 63) * 4, ar = ((lane + 1) & 63) * 4;
    const int chunk = pair / p.nstrips;                 // within its segment
    const int strip = pair - chunk * p.nstrips;
    const int n0 = p.n0, n1 = p.n1;
    const int chunk_rows = second ? p.seg1_chunk_rows : p.chunk_rows;
    const int r0 = (second ? p.seg1_begin : p.seg0_begin) + chunk * chunk_rows;
    const int r1 = min(r0 + chunk_rows, second ? p.seg1_end : p.seg0_end);
    const int nrows = r1 - r0;                         // >= 1 by construction of the grid

    const int col = strip * FGROUP - FHALO + pp * FPITCH + lane;
    int jc = col;
    if (p.bc1 == 1) jc = jc < 0 ? jc + n1 : (jc >= n1 ? jc - n1 : jc);
    jc = min(max(jc, 0), n1 - 1);
    const unsigned jc8 = (unsigned) jc * 8u;

    const long row_stride = p.row_stride;
    const double* in = p.u_in;
    const typename A::Gamma gl = A::gamma_law(p.gamma);
    const typename A::Limiter lim = A::limiter(p.theta);
    const uint32_t n1u = (uint32_t) n1, colu = (uint32_t) col;
    StatusAcc acc;

    // ================================================================ FIRST STAGE over rows a0 .. a0 + T - 1 of a strip ==========================
    // The row loop of euler2d.hip on the step-start field; each row of u1 goes to the hand-off ring instead of memory. Run by the PRODUCER over
    // rows r0 - 2 .. r1 + 1 with a barrier per row - or (round 5 probe, MH_FUSED_SPLIT_LEAD = 1) over r0 .. r1 + 1, with the CONSUMER forming
    // the two rows r0 - 2, r0 - 1 itself before it joins the barriers (same functions on the same values: same bits; not taken, see below).
    // slot0: ring slot of row a0 (rows are numbered from r0 - 2 in both rings); BARRIERS: the producer's form.
    // step-start rows beyond the field: the periodic image, or (outflow) the edge row - what the stored ghost rows hold, two rows
    // further out than they reach
    const int bc0_lo = p.bc0_lo, bc0_hi = p.bc0_hi;
    auto row_of = [in, row_stride, n0, bc0_lo, bc0_hi] (int r)
    {
        // (the row loop requests rows up to two beyond the last one it uses: EXTERNAL sides stop at the four rows that exist)
        int m = r;
#ifdef MH_PROBE_FUSED_NO_EXTERNAL_CLAMP      // the round-3 fault, rebuilt for tests/test_gpu_row_range.py (check builds only: MH_ROW holds the access to the rows that exist)
        if (r < 0)        m = bc0_lo == 1 ? r + n0 : (bc0_lo == 2 ? r : 0);
        else if (r >= n0) m = bc0_hi == 1 ? r - n0 : (bc0_hi == 2 ? r : n0 - 1);
#else
        if (r < 0)        m = bc0_lo == 1 ? r + n0 : (bc0_lo == 2 ? max(r, -4) : 0);
        else if (r >= n0) m = bc0_hi == 1 ? r - n0 : (bc0_hi == 2 ? min(r, n0 + 3) : n0 - 1);
#endif
        return in + (long) (MH_ROW(m, bc0_lo == 2 ? -4 : -2, bc0_hi == 2 ? n0 + 3 : n0 + 1) + 2) * row_stride;          // (EXTERNAL: rows -4, -3 and n0 + 2, n0 + 3 lie outside the stored ghost rows: the slab stepper allocates them)
    };
    auto first_stage = [&] (const int a0, const int T, const int slot0, const int last_needed, auto barriers) __attribute__((always_inline))
    {
        constexpr bool BARRIERS = decltype(barriers)::value;
        (void) last_needed;                             // (the noclamp probe build does not use it)
        // slot of step-start row x: (x - (r0 - 2)) mod USLOTS
        auto ring_put = [&] (int slot, const State5& raw)
        {
#pragma unroll
            for (int q = 0; q < 5; ++q) if (live(q)) start_rows[slot][vi(q)][lane] = raw[q];
        };
        auto ring_get = [&] (int slot) -> State5
        {
            State5 Uq;
#pragma unroll
            for (int q = 0; q < 5; ++q) Uq[q] = live(q) ? start_rows[slot][vi(q)][lane] : 0.0;
            return Uq;
        };
        const bool real_col = lane >= 2 && lane < FWAVE - 2 && col >= 0 && col < n1;     // a cell of the grid whose first-stage value is valid here
        const bool works = ! MH_FUSED_MASK_HALO || (lane >= 2 && lane < FWAVE - 2);

        State5 U[3], P[3], G[3], Fx[3], D[3];
        {
            const State5 Pa = A::c2p(load_row<PLANAR>(row_of(a0 - 2), p.plane_stride, jc8), gl);
            const State5 Pb = A::c2p(load_row<PLANAR>(row_of(a0 - 1), p.plane_stride, jc8), gl);
            U[0] = load_row<PLANAR>(row_of(a0), p.plane_stride, jc8);
            U[1] = load_row<PLANAR>(row_of(a0 + 1), p.plane_stride, jc8);
            U[2] = load_row<PLANAR>(row_of(min(a0 + 2, last_needed)), p.plane_stride, jc8);
            P[0] = A::c2p(U[0], gl);
            P[1] = A::c2p(U[1], gl);
            ring_put(slot0 % USLOTS, U[0]);
            ring_put((slot0 + 1) % USLOTS, U[1]);
            U[0] = load_row<PLANAR>(row_of(min(a0 + 3, last_needed)), p.plane_stride, jc8);
            const State5 Dab = A::difference(Pa, Pb), Db0 = A::difference(Pb, P[0]);
            D[0] = A::difference(P[0], P[1]);
            const State5 Gb = A::plm_from_differences(Dab, Db0, lim);
            G[0] = A::plm_from_differences(Db0, D[0], lim);
            Fx[0] = A::template flux<RIEMANN, 0>(A::plus(Pb, Gb, lim), A::minus(P[0], G[0], lim), gl);
        }
        if (__any(!(P[0][4] >= 0.0) || !(P[1][4] >= 0.0)))
        {
            if (real_col && !(P[0][4] >= 0.0) && a0 >= 0 && a0 < n0) acc.note_value(P[0][4], MH_STATUS_NEG_PRESSURE, (uint32_t) a0 * n1u + colu);
            if (real_col && !(P[1][4] >= 0.0) && a0 + 1 >= 0 && a0 + 1 < n0) acc.note_value(P[1][4], MH_STATUS_NEG_PRESSURE, (uint32_t) (a0 + 1) * n1u + colu);
        }

        auto row_step = [&] (int a, int t, auto k0) __attribute__((always_inline))
        {
            constexpr int K0 = decltype(k0)::value, K1 = (K0 + 1) % 3, K2 = (K0 + 2) % 3;
            // (the look-ahead stops at the last row this run uses: the rows beyond it would be read for nothing - and, for a slab's
            // interior launch next to a cut, while the exchange on the side stream may still be writing them)
#ifdef MH_PROBE_FUSED_NO_EXTERNAL_CLAMP      // round 3's kernel for tests/test_gpu_row_range.py: look-ahead two rows beyond the last one used, no clamp in row_of
            U[K1] = load_row<PLANAR>(row_of(a + 4), p.plane_stride, jc8);
#else
            U[K1] = load_row<PLANAR>(row_of(min(a + 4, last_needed)), p.plane_stride, jc8);
#endif
            P[K2] = A::c2p(U[K2], gl);
            if constexpr (BARRIERS) ring_put((slot0 + t + 2) % USLOTS, U[K2]);          // (the consumer's two rows wait in the slots its prologue filled)
            const bool bad_pressure = !(P[K2][4] >= 0.0);
            D[K1] = A::difference(P[K1], P[K2]);
            G[K1] = A::plm_from_differences(D[K0], D[K1], lim);
            if (works) Fx[K1] = A::template flux<RIEMANN, 0>(A::plus(P[K0], G[K0], lim), A::minus(P[K1], G[K1], lim), gl);

            const State5 Dr = A::difference(P[K0], lane_from<1, PLANAR>(P[K0], al, ar, false));
            const State5 Gy = A::plm_from_differences(lane_from<2, PLANAR>(Dr, al, ar, true), Dr, lim);
            const State5 SL = lane_from<4, PLANAR>(A::plus(P[K0], Gy, lim), al, ar, true);
            const State5 Fy_lo = A::template flux<RIEMANN, 1>(SL, A::minus(P[K0], Gy, lim), gl);
            const State5 Fy_hi = lane_from<8, PLANAR>(Fy_lo, al, ar, false);

            const State5 Uc = ring_get((slot0 + t) % USLOTS);
            State5 Un = Uc;
            if (works)
            {
#pragma unroll
                for (int q = 0; q < 5; ++q) if (live(q)) Un[q] = A::update2(Uc[q], Fx[K0][q], Fx[K1][q], Fy_lo[q], Fy_hi[q], p.cx, p.cy);
            }
            const bool bad_density = !(Un[0] > 0.0);
            if (__any(bad_pressure || bad_density))
            {
                if (real_col && bad_pressure && BARRIERS && a + 2 >= 0 && a + 2 < n0) acc.note_value(P[K2][4], MH_STATUS_NEG_PRESSURE, (uint32_t) (a + 2) * n1u + colu);
                if (real_col && bad_density && a >= 0 && a < n0) acc.note_value(Un[0], MH_STATUS_NEG_DENSITY, (uint32_t) a * n1u + colu);
            }
            const int slot = (slot0 + t) % FSLOTS;
#pragma unroll
            for (int q = 0; q < 5; ++q) if (live(q)) hand[slot][vi(q)][lane] = Un[q];
            if constexpr (BARRIERS) pair_barrier();          // barrier #t: row a is in the ring
        };

        int t = 0;
        for (; t + 3 <= T; t += 3)
        {
            row_step(a0 + t, t, std::integral_constant<int, 0>());
            row_step(a0 + t + 1, t + 1, std::integral_constant<int, 1>());
            row_step(a0 + t + 2, t + 2, std::integral_constant<int, 2>());
        }
        if (t < T) row_step(a0 + t, t, std::integral_constant<int, 0>());
        if (t + 1 < T) row_step(a0 + t + 1, t + 1, std::integral_constant<int, 1>());
    };
    // MH_FUSED_SPLIT_LEAD = 1: the consumer forms first-stage rows r0 - 2, r0 - 1 itself; 0 (default): the producer forms all of r0 - 2 .. r1 + 1.
    // Built, bit-identical (tests/test_gpu_fused_rk2.py ran green on it) and NOT taken: 76.6 / 140.1 / 530.8 against 76.4 / 138.9 / 532.5 us per
    // step at 512 / 1024 / 4096 rows (profiles/r05/ab_split_lead.jsonl). The chunk's fill is WORK, not a wave waiting: the hardware already
    // places producers and consumers of different workgroups on one SIMD (scripts/probes/wave_placement.hip: every SIMD holds one or two
    // producers of its three waves), so the lead rows run in the issue slots the waiting consumers leave - splitting them moves nothing.
#ifndef MH_FUSED_SPLIT_LEAD
#define MH_FUSED_SPLIT_LEAD 0
#endif
    constexpr int LEAD = MH_FUSED_SPLIT_LEAD ? 2 : 4;      // barriers the consumer passes before its prologue = rows the producer forms ahead of it

    if (role == 0)
    {
        // ================================================================ PRODUCER: first stage, rows r1 + 2 - (nrows + LEAD) .. r1 + 1 =========
        // (first-stage rows r0 - 2 .. r1 + 1 need step-start rows r0 - 4 .. r1 + 3)
        first_stage(r0 - 2 + (4 - LEAD), nrows + LEAD, 4 - LEAD, r1 + 3, std::true_type());
    }
    else
    {
        // ================================================================ CONSUMER: second stage + RK average, rows r0 .. r1 - 1 ========
        const int out_lo = pp > 0 ? 2 : FHALO, out_hi = pp < FPAIRS - 1 ? FWAVE - 2 : FWAVE - FHALO;
        const bool writes = lane >= out_lo && lane < out_hi && col < n1;
        const bool works = ! MH_FUSED_MASK_HALO || (lane >= out_lo && lane < out_hi);
        const unsigned col8 = (unsigned) (writes ? col : 0) * 8u;
        // ghost columns of the first-stage field: outflow = the edge column's value, i.e. another lane's entry of the ring
        int src_lane = lane;
        if (p.bc1 != 1) src_lane = lane + (min(max(col, 0), n1 - 1) - col);
        // the neighbouring pair's ring where this pair's producer has no valid value for the column (its lanes 0, 1, 62, 63)
        int other = pp, src_lane_other = src_lane;
        if (src_lane < 2 && pp > 0) { other = pp - 1; src_lane_other = src_lane + FPITCH; }
        else if (src_lane > FWAVE - 3 && pp < FPAIRS - 1) { other = pp + 1; src_lane_other = src_lane - FPITCH; }
        const bool from_other = other != pp && src_lane_other >= 2 && src_lane_other <= FWAVE - 3;
        src_lane = min(max(src_lane, 0), FWAVE - 1);
        // one LDS read per value either way: the lane's offset into the rings of all pairs, [pair][slot][variable][lane] flattened
        const double* const hand_flat = &hand_all[0][0][0][0];
        const int hand_off = from_other ? other * (FSLOTS * NV * FWAVE) + src_lane_other : pp * (FSLOTS * NV * FWAVE) + src_lane;
        // first-stage row rr as the producer left it; outflow ghost rows are the edge rows' slots (periodic and EXTERNAL: the producer
        // worked on the wrapped row / on the neighbour's rows)
        auto hand_row = [&] (int rr) -> State5
        {
            const int m = rr < 0 ? (bc0_lo == 0 ? 0 : rr) : (rr >= n0 ? (bc0_hi == 0 ? n0 - 1 : rr) : rr);
            const int slot = (m - (r0 - 2)) % FSLOTS;
            State5 Uq;
#pragma unroll
            for (int q = 0; q < 5; ++q) Uq[q] = live(q) ? hand_flat[hand_off + (slot * NV + vi(q)) * FWAVE] : 0.0;
            return Uq;
        };
        auto row_off = [row_stride, n0] (int r) { (void) n0; return (long) (MH_ROW(r, -2, n0 + 1) + 2) * row_stride; };

        if constexpr (LEAD == 2)
        {
            // first-stage rows r0 - 2, r0 - 1 by this wave (slots 0, 1 of both rings; the producer starts at row r0, slot 2), from step-start rows
            // r0 - 4 .. r0 + 1; then the producer's rows r0, r0 + 1
            first_stage(r0 - 2, 2, 0, r0 + 1, std::false_type());
            pair_barrier(); pair_barrier();                                          // barriers #0, #1: rows r0, r0 + 1 are in the ring
        }
        else
        {
            pair_barrier(); pair_barrier(); pair_barrier(); pair_barrier();          // barriers #0..#3: rows r0 - 2 .. r0 + 1 are in the ring
        }

        State5 P[3], G[3], Fx[3], D[3];
        {
            const State5 Pa = A::c2p(hand_row(r0 - 2), gl);
            const State5 Pb = A::c2p(hand_row(r0 - 1), gl);
            P[0] = A::c2p(hand_row(r0), gl);
            P[1] = A::c2p(hand_row(r0 + 1), gl);
            const State5 Dab = A::difference(Pa, Pb), Db0 = A::difference(Pb, P[0]);
            D[0] = A::difference(P[0], P[1]);
            const State5 Gb = A::plm_from_differences(Dab, Db0, lim);
            G[0] = A::plm_from_differences(Db0, D[0], lim);
            Fx[0] = A::template flux<RIEMANN, 0>(A::plus(Pb, Gb, lim), A::minus(P[0], G[0], lim), gl);
        }
        if (__any(!(P[0][4] >= 0.0) || !(P[1][4] >= 0.0)))
        {
            if (writes && !(P[0][4] >= 0.0)) acc.note_value(P[0][4], MH_STATUS_NEG_PRESSURE, (uint32_t) r0 * n1u + colu);
            if (writes && !(P[1][4] >= 0.0) && r0 + 1 < n0) acc.note_value(P[1][4], MH_STATUS_NEG_PRESSURE, (uint32_t) (r0 + 1) * n1u + colu);
        }

        auto row_step = [&] (int r, auto k0) __attribute__((always_inline))
        {
            constexpr int K0 = decltype(k0)::value, K1 = (K0 + 1) % 3, K2 = (K0 + 2) % 3;
            pair_barrier();                                  // barrier #(r - r0 + LEAD): row r + 2 is in the ring
            State5 Ubase;                                    // the step-start row, for the average: the producer kept it (slot of row r)
            {
                const int slot = (r - (r0 - 2)) % USLOTS;
#pragma unroll
                for (int q = 0; q < 5; ++q) Ubase[q] = live(q) ? start_rows[slot][vi(q)][lane] : 0.0;
            }
            P[K2] = A::c2p(hand_row(r + 2), gl);
            const bool bad_pressure = !(P[K2][4] >= 0.0);
            D[K1] = A::difference(P[K1], P[K2]);
            G[K1] = A::plm_from_differences(D[K0], D[K1], lim);
            if (works) Fx[K1] = A::template flux<RIEMANN, 0>(A::plus(P[K0], G[K0], lim), A::minus(P[K1], G[K1], lim), gl);

            const State5 Dr = A::difference(P[K0], lane_from<1, PLANAR>(P[K0], al, ar, false));
            const State5 Gy = A::plm_from_differences(lane_from<2, PLANAR>(Dr, al, ar, true), Dr, lim);
            const State5 SL = lane_from<4, PLANAR>(A::plus(P[K0], Gy, lim), al, ar, true);
            const State5 Fy_lo = A::template flux<RIEMANN, 1>(SL, A::minus(P[K0], Gy, lim), gl);
            const State5 Fy_hi = lane_from<8, PLANAR>(Fy_lo, al, ar, false);

            const State5 Uc = hand_row(r);
            State5 Un = Uc;
            if (works)
            {
#pragma unroll
                for (int q = 0; q < 5; ++q) if (live(q)) Un[q] = A::combine(Ubase[q], A::update2(Uc[q], Fx[K0][q], Fx[K1][q], Fy_lo[q], Fy_hi[q], p.cx, p.cy), 0.5);
            }
            const bool bad_density = !(Un[0] > 0.0);
            if (__any(bad_pressure || bad_density))
            {
                if (writes && bad_pressure && r + 2 < n0) acc.note_value(P[K2][4], MH_STATUS_NEG_PRESSURE, (uint32_t) (r + 2) * n1u + colu);
                if (writes && bad_density) acc.note_value(Un[0], MH_STATUS_NEG_DENSITY, (uint32_t) r * n1u + colu);
            }
            if (writes)
            {
                store_row(p.u_out + row_off(r), p.plane_stride, col8, Un);
                if (r < 2 || r >= n0 - 2)          // the stored ghost rows of the result (edge rows only: wave-uniform, cold)
                {
                    if (bc0_lo == 0 && r == 0) { store_row(p.u_out + row_off(-1), p.plane_stride, col8, Un); store_row(p.u_out + row_off(-2), p.plane_stride, col8, Un); }
                    if (bc0_hi == 1 && r < 2) store_row(p.u_out + row_off(n0 + r), p.plane_stride, col8, Un);
                    if (bc0_hi == 0 && r == n0 - 1) { store_row(p.u_out + row_off(n0), p.plane_stride, col8, Un); store_row(p.u_out + row_off(n0 + 1), p.plane_stride, col8, Un); }
                    if (bc0_lo == 1 && r >= n0 - 2) store_row(p.u_out + row_off(r - n0), p.plane_stride, col8, Un);
                }
            }
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
    }
    acc.commit(p.status);
}

// how the last launch of this translation unit cut its rows: {chunk rows, chunk rows of the second segment, chunks of the first, chunks} (tests)
static int last_cut[4] = {0, 0, 0, 0};
void euler2d_fused_last_cut(int out[4]) { for (int k = 0; k < 4; ++k) out[k] = last_cut[k]; }

// with_cuts: MH_BC_EXTERNAL sides are accepted too - the caller (slab.hip) keeps FOUR rows of the neighbour beyond such a side
bool euler2d_fused_rk2_available(const mh_euler_cart_desc* d, bool with_cuts)
{
    auto side_ok = [with_cuts] (int bc) { return bc == MH_BC_OUTFLOW || bc == MH_BC_PERIODIC || (with_cuts && bc == MH_BC_EXTERNAL); };
    return d->rank == 2 && d->arith == MH_ARITH_FAST && d->plm_theta >= 0.0 && d->n[0] >= 8 && d->n[1] >= 8
        && side_ok(d->bc_lo0) && side_ok(d->bc_hi0) && ((d->bc_lo0 == MH_BC_PERIODIC) == (d->bc_hi0 == MH_BC_PERIODIC))
        && (d->bc_transverse == MH_BC_OUTFLOW || d->bc_transverse == MH_BC_PERIODIC);
}

// u_out = u_in * 0.5 + advance(advance(u_in)) * 0.5 over the whole field (both with stored ghost rows, layout of include/mara_hip.h); the two
// fields must differ. chunk_rows: the descriptor's, or the default below.
hipError_t euler2d_fused_rk2_launch(const mh_euler_cart_desc* d, const double* u_in, double* u_out, double dt, int32_t* status, hipStream_t stream,
                                    LaunchEvents ev, bool with_cuts)
{
    return euler2d_fused_rk2_launch_rows(d, u_in, u_out, dt, 0, d->n[0], 0, 0, status, stream, ev, with_cuts, 0);
}

// ... over rows [a, b) and, in the same launch, [a2, b2) (b2 <= a2: none) of the field
// late_blocks: workgroups of ANOTHER launch that hold slots of the chip when this one starts (a slab's edge launch beside its interior), see below
hipError_t euler2d_fused_rk2_launch_rows(const mh_euler_cart_desc* d, const double* u_in, double* u_out, double dt, int a, int b, int a2, int b2,
                                         int32_t* status, hipStream_t stream, LaunchEvents ev, bool with_cuts, int late_blocks)
{
    if (! euler2d_fused_rk2_available(d, with_cuts) || u_in == u_out) return hipErrorInvalidValue;
    if (a < 0 || b > d->n[0] || b <= a || (b2 > a2 && (a2 < b || b2 > d->n[0]))) return hipErrorInvalidValue;
    int rows0 = b - a, rows1 = b2 > a2 ? b2 - a2 : 0;
    const int longest = rows0 > rows1 ? rows0 : rows1;
    Fused2dParams p;
    p.u_in = u_in; p.u_out = u_out; p.status = status;
    p.n0 = d->n[0]; p.n1 = d->n[1];
    p.plane_stride = p.n1;
    p.row_stride = 5L * p.n1;
    p.nstrips = (p.n1 + FGROUP - 1) / FGROUP;
    int rounds = 1;          // residency rounds of the launch at the default chunk length
    if (d->chunk_rows > 0) p.chunk_rows = d->chunk_rows;
    else
    {
        // A pair pays eight pipeline-fill rows per chunk, so chunks are long; and 1024 pairs are resident at a time (four per CU), so the
        // launch takes ceil(pairs / 1024) residency rounds of about (chunk + 8) rows each: the chunk is the shortest one that fills R rounds
        // to the brim, for the smallest R that keeps it near 100 rows. Measured at 4096^2, 74 strips (profiles/r03/ab_fused_chunks.jsonl):
        // 100 rows (41 chunks, 2.96 rounds) 0.631-0.638 ms per step; 106 rows (2.82 rounds) 0.658; 75-79 rows (3.8-4 rounds) 0.645;
        // 64 rows 0.669; 152 rows (1.95 rounds) 0.651; 316 rows (one round) 0.680; two launches 0.669-0.697 on the same boxes.
        // (a slab's interior launch does NOT leave room for its edge launch's pairs: measured slower, 125 against 112 us per step at 512 rows)
        const int resident = device_cu_count() * 4 * fused_waves_per_simd(d->planar > 0) / (2 * FPAIRS);          // workgroups on the chip at a time
        auto chunk_for = [&] (int r) { const int nch = resident * r / p.nstrips > 0 ? resident * r / p.nstrips : 1; return (rows0 + rows1 + nch - 1) / nch; };
        while (chunk_for(rounds) > (d->planar > 0 && MH_FUSED_PLANAR_WAVES >= 3 ? 80 : 112)) ++rounds;
        p.chunk_rows = chunk_for(rounds);
        if (p.chunk_rows < 8) p.chunk_rows = 8;
    }
    if (p.chunk_rows > longest) p.chunk_rows = longest;
    p.seg1_chunk_rows = p.chunk_rows;
    // TAPER (round 5): a one-round launch fills the chip's workgroup slots to the brim, so when `late_blocks` slots are held by another launch at
    // its start (the edge strips of a slab, issued first on the high-priority stream), as many of THIS launch's workgroups - the ones launched
    // last - start only when those end, `taper` row-times later, and end that much after all the others: 156.6 instead of 134.8 us per step
    // for a 1024-row slab with neighbours (profiles/r05/thin_slab_scaling.md). The rows are therefore cut into long chunks, launched first, and
    // ceil(late_blocks / nstrips) SHORTER chunks per strip, launched last: c_short = c_long - taper, so that all end together. Results do
    // not depend on the cut (tests/test_gpu_fused_rk2.py, tests/test_gpu_slab_group.py). MH_FUSED_TAPER_ROWS: the head start lost, in rows
    // (default 8 = an edge chunk's four rows and its fill; 0 = off).
    // (read per launch, not cached: tests/test_gpu_slab_group.py and test_gpu_cloud_fused.py exercise the tapered cut on small grids with
    // MH_FUSED_TAPER_MIN - the shortest short chunk for which the taper applies, default 24 rows)
    const int taper = [] { const char* v = getenv("MH_FUSED_TAPER_ROWS"); return v ? atoi(v) : 8; } ();
    const int taper_min = [] { const char* v = getenv("MH_FUSED_TAPER_MIN"); return v ? atoi(v) : 24; } ();
    if (late_blocks > 0 && taper > 0 && rows1 == 0 && d->chunk_rows <= 0)
    {
        const int resident = device_cu_count() * 4 * fused_waves_per_simd(d->planar > 0) / (2 * FPAIRS);
        const int nch = resident * rounds / p.nstrips;                         // chunks per strip of the launch's residency rounds (the workgroups launched
                                                                               // last are the last round's late starters whatever the number of rounds)
        const int nshort = (late_blocks + p.nstrips - 1) / p.nstrips;
        const int clong = nch > 0 ? (rows0 + nshort * taper + nch - 1) / nch : 0;
        // one round, and chunks long enough for the taper to pay: measured with the exchange to self (profiles/r05/ab_taper.txt, us per step, taper
        // 0 / 6 / 8 / 10 / 12 rows): 1024 rows (50-row chunks) 154.5 - 156.9 / 146.8 - 149.4 / 147.0 - 150.7 / 145.3 - 150.1 / 145.7 - 149.7;
        // 512 rows (24-row chunks, which fit the 21 chunks per strip exactly) 80.4 - 82.0 / 81.3 - 81.9 / 81.8 - 82.9 / 81.2 - 82.1 / 83.8 - 84.3: not there
        if (nch > nshort && clong - taper >= taper_min && clong <= (d->planar > 0 && MH_FUSED_PLANAR_WAVES >= 3 ? 80 : 112))
        {
            const int long_rows = (nch - nshort) * clong;
            if (long_rows < rows0)
            {
                p.chunk_rows = clong;
                p.seg1_chunk_rows = clong - taper;
                a2 = a + long_rows; b2 = b;                                    // the short chunks take the END of the range
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
    auto side = [] (int bc) { return bc == MH_BC_PERIODIC ? 1 : (bc == MH_BC_EXTERNAL ? 2 : 0); };
    p.bc0_lo = side(d->bc_lo0);
    p.bc0_hi = side(d->bc_hi0);
    p.bc1 = d->bc_transverse == MH_BC_PERIODIC ? 1 : 0;
    p.gamma = d->gamma; p.theta = d->plm_theta;
    p.cx = dt / d->dl[0]; p.cy = dt / d->dl[1];
    const dim3 grid(p.nstrips * p.nchunks), block(2 * FWAVE * FPAIRS);
    // d->planar > 0: the caller (a stepper that has verified it at upload) knows the field's third momentum to be identically zero
#define MH_FUSED_LAUNCH(R, PL) do { if (ev.stop) hipExtLaunchKernelGGL((euler2d_fused_rk2_kernel<R, PL>), grid, block, 0, stream, ev.start, ev.stop, 0, p); \
                                    else         hipLaunchKernelGGL((euler2d_fused_rk2_kernel<R, PL>), grid, block, 0, stream, p); } while (0)
    if (d->riemann == MH_RIEMANN_HLLC) { if (d->planar > 0) MH_FUSED_LAUNCH(1, true); else MH_FUSED_LAUNCH(1, false); }
    else                               { if (d->planar > 0) MH_FUSED_LAUNCH(0, true); else MH_FUSED_LAUNCH(0, false); }
#undef MH_FUSED_LAUNCH
    return hipGetLastError();
}

int euler2d_fused_rk2_blocks_per_chunk(const mh_euler_cart_desc* d) { return (d->n[1] + FGROUP - 1) / FGROUP; }

// row-range guard (row_check.hpp): what this translation unit's kernels asked for; false in product builds
MH_ROW_RANGE_READER(rows_requested_euler2d_fused)

} // namespace mh
