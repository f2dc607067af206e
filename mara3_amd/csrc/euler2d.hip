// 2-D uniform-cartesian Euler Runge-Kutta stage for gfx950 (MI355X).
//
// Replaces one evaluation of the lazily-composed `advance` expression
// (src/subprog_cloud.cpp:511-584, specialised to mara::euler on a cartesian
// grid; see oracle/ref_drivers/euler_cart_ref.cpp for the reference-header
// form) plus, when stage_weight != 1, the RK combine of
// src/subprog_cloud.cpp:682-695.
//
// Design (HBM-bound stencil with ~400 (fast) to ~900 (strict) issue slots per cell-row, no MFMA):
//  * One 64-lane wavefront owns a strip of 60 columns (+2 halo lanes on each
//    side) and MARCHES along axis 0 over `chunk_rows` rows. Lanes run along
//    axis 1, the contiguous axis, so every plane access is one coalesced 512 B
//    row segment.
//  * Along the march direction everything a cell needs lives in the lane's
//    registers: primitives of rows r..r+2, the PLM slope of row r and the
//    flux through face r-1/2, which is reused from the previous iteration -
//    each axis-0 face flux is computed exactly once.
//  * Along axis 1 the neighbours' primitives, face states and fluxes move
//    between lanes with DPP wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1):
//    each axis-1 face flux is also computed once, by the lane on its right,
//    and handed to the lane on its left.
//  * Only the conserved planes are read (once, + 4/chunk_rows and 4/60 halo
//    re-reads that hit in L2) and written; primitives, slopes and fluxes never
//    touch memory. The second RK2 stage also streams the step-start field
//    and writes the averaged result in place.
//  * Boundary conditions need no branches in the row loop: axis-1 ghosts are
//    clamped / wrapped column indices computed once per wave; axis-0 ghosts are
//    two stored rows per side, refreshed by whichever wave writes the edge rows.
//  * Launch order: workgroups are dealt round-robin over the 8 XCDs, so the
//    work-item order keeps neighbouring tiles on one XCD (halo re-reads hit its
//    L2); a large launch ends on short waves (graded tail) in every XCD's share.
//  * Arithmetic is a template policy: StrictArith (bit-identical to the
//    reference) or FastArith (euler_device_fast.hpp; there the register window
//    holds primitives only and limiter differences are shared between cells).
//
// Algorithmic HBM bytes per cell per stage: 80 (first stage) / 120 (second
// stage of RK2) => 200 B per zone-update for RK2.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "euler_device.hpp"
#include "euler_device_fast.hpp"
#include "launch.hpp"
#include "status_device.hpp"
#include "euler2d_rows.hpp"
#include "row_check.hpp"

namespace mh {

static constexpr int WAVE = 64;
static constexpr int HALO = 2;
static constexpr int STRIP = WAVE - 2 * HALO;   // 60 output columns per wave
static constexpr int WAVES_PER_BLOCK = 4;
#ifndef MH_MIN_WAVES
#define MH_MIN_WAVES 2
#endif

struct Stage2dParams
{
    const double* u_in;
    const double* u_base;
    double*       u_out;
    int32_t*      status;
    long   plane_stride;     // doubles between planes of one row
    long   row_stride;       // doubles between consecutive rows
    int    n0, n1;           // local rows, columns
    int    row_begin, row_end;
    int    row_begin2, row_end2, nchunks_a;   // optional second row range: chunks >= nchunks_a march [row_begin2, row_end2)
    int    chunk_rows;
    int    chunk_rows2;                       // rows per wave in the second range (a graded tail: shorter waves at the end of the launch)
    int    tail_blocks_per_xcd;               // workgroups of each XCD's share that lie in the second range (0: plain XCD-aware order)
    int    nstrips, nchunks;
    int    bc_lo0, bc_hi0, bc1;
    double gamma, theta, cx, cy, weight;
};

// The work of one workgroup `b` of a stage launch of `nblocks` workgroups (the kernel below is a thin wrapper). Kept separate from the
// kernel because a "mixed" launch was measured that dealt first-stage and second-stage work items of two row bands to neighbouring
// workgroups of ONE grid, hoping that the issue-bound and the bandwidth-bound kind would overlap on every CU: 0.730 against 0.737 ms per
// 4096^2 step (FAST, HLLC), slower for every other variant (profiles/r02/mixed_launch_probe.jsonl) - not kept.
template<class A, int RIEMANN, bool PLM, bool COMBINE>
__device__ __forceinline__ void stage_body(const Stage2dParams& p, int b, const int nblocks)
{
    // ---- which (chunk, strip) does this wave own? XCD-aware: consecutive work
    // items (neighbouring strips, then neighbouring chunks) go to the same XCD so
    // that halo re-reads hit in that XCD's L2 (blocks are dealt round-robin over 8 XCDs).
    {
        // blocks are dealt round-robin over the 8 XCDs: XCD x takes blockIdx = x, x + 8, ... in order. Its share is a contiguous run of
        // body work items followed by a contiguous run of tail items (the short waves of the graded tail), so that every XCD ends on short waves.
        const int per_xcd = nblocks >> 3, tail = p.tail_blocks_per_xcd, body = per_xcd - tail;
        if (b < per_xcd * 8)
        {
            const int x = b & 7, s = b >> 3;
            b = s < body ? x * body + s : 8 * body + x * tail + (s - body);
        }
    }
    // the wave index is uniform across the wave: tell the compiler, so that everything derived from it (chunk,
    // strip, row loop, row addresses) lives in scalar registers and costs no vector instructions
    const int w = __builtin_amdgcn_readfirstlane(b * WAVES_PER_BLOCK + (int) (threadIdx.x >> 6));
    if (w >= p.nstrips * p.nchunks) return;
    const int lane = threadIdx.x & 63;
    const int chunk = w / p.nstrips;
    const int strip = w - chunk * p.nstrips;

    int r0 = p.row_begin + chunk * p.chunk_rows;
    int r1 = min(r0 + p.chunk_rows, p.row_end);
    if (chunk >= p.nchunks_a)
    {
        r0 = p.row_begin2 + (chunk - p.nchunks_a) * p.chunk_rows2;
        r1 = min(r0 + p.chunk_rows2, p.row_end2);
    }

    // ---- column of this lane, with the axis-1 boundary condition folded into the index
    const int col = strip * STRIP - HALO + lane;
    int jc = col;
    if (p.bc1 == 1) { jc = jc < 0 ? jc + p.n1 : (jc >= p.n1 ? jc - p.n1 : jc); jc = min(max(jc, 0), p.n1 - 1); }
    else            { jc = min(max(jc, 0), p.n1 - 1); }
    const bool writes = lane >= HALO && lane < WAVE - HALO && col < p.n1;

    const long row_stride = p.row_stride;
    const double* in = p.u_in;                       // row r, plane q lives at (r + 2) * row_stride + q * plane_stride
    const int rows_hi = p.n0 + 1;                     // the rows that exist: -2 .. n0 + 1 (row_check.hpp)
    auto row_off = [row_stride, rows_hi] (int r) { (void) rows_hi; return (long) (MH_ROW(r, -HALO, rows_hi) + HALO) * row_stride; };
    const unsigned jc8 = (unsigned) jc * 8u, col8 = (unsigned) (writes ? col : 0) * 8u;

    // A::planar (StrictArithT<true>, FastArithT<true>): the field's third momentum is identically zero (+0.0 bit for bit in STRICT) - not read, not
    // exchanged, not computed, written as +0.0 (mh_euler_cart_desc.planar; euler_device.hpp says why the other four components keep their bits)
    constexpr bool PL = A::planar;
    constexpr auto live = [] (int q) { return ! (PL && q == 3); };
    const double gamma = p.gamma, theta = p.theta;
    const typename A::Gamma gl = A::gamma_law(gamma);
    const typename A::Limiter lim = A::limiter(theta);

    // MH_ARITH_FAST keeps only primitives in the register window. The conserved values of rows r, r+1, r+2 wait for their update in a
    // per-wave LDS ring (private to the wave: no barrier) - so that the update starts from the STORED state and the scheme conserves to
    // rounding, as the reference does (re-forming the state from the primitives, A::p2c, cost an ulp of energy per cell and step)
    constexpr bool lds_ring = A::recompute_conserved && A::lds_conserved_ring;
    __shared__ double own_rows[lds_ring ? WAVES_PER_BLOCK : 1][3][5][WAVE];
    const int wave_in_block = (int) (threadIdx.x >> 6);
    auto ring_put = [&] (int slot, const State5& raw)
    {
        if constexpr (lds_ring)
        {
#pragma unroll
            for (int q = 0; q < 5; ++q) if (live(q)) own_rows[wave_in_block][slot][q][lane] = raw[q];
        }
    };
    auto ring_get = [&] (int slot) -> State5
    {
        State5 Uq;
#pragma unroll
        for (int q = 0; q < 5; ++q) Uq[q] = live(q) ? own_rows[lds_ring ? wave_in_block : 0][slot][q][lane] : 0.0;
        return Uq;
    };

    // ---- register window: three slots used as rings (index = row mod 3 relative to the chunk start), so that the
    // row loop, unrolled by three, needs no register-to-register rotation at all.
    //   U[k]: conserved of rows r, r+1, r+2        P[k]: primitives of rows r, r+1, r+2
    //   G[k]: axis-0 slope of rows r, r+1          Fx[k]: axis-0 flux through faces r-1/2, r+1/2
    //   D[k]: (FAST) P of row r+1 - P of row r, the limiter's one-sided difference across face r+1/2
    State5 U[3], P[3], G[3], Fx[3], D[3];
    {
        const State5 Pa = A::c2p(load_row<PL>(in + row_off(r0 - 2), p.plane_stride, jc8), gl);
        const State5 Pb = A::c2p(load_row<PL>(in + row_off(r0 - 1), p.plane_stride, jc8), gl);
        U[0] = load_row<PL>(in + row_off(r0), p.plane_stride, jc8);
        U[1] = load_row<PL>(in + row_off(r0 + 1), p.plane_stride, jc8);
        U[2] = load_row<PL>(in + row_off(r0 + 2), p.plane_stride, jc8);   // first prefetch
        P[0] = A::c2p(U[0], gl);
        P[1] = A::c2p(U[1], gl);
        ring_put(0, U[0]);
        ring_put(1, U[1]);
        // (recompute_conserved: U[] is the ring of LOADED rows instead - slot (row - r0) mod 3 holds row r+2, r+3 or r+4 until its
        // conversion; rows r0, r0+1 are converted already and their slots take rows r0+3, r0+4)
        if constexpr (A::recompute_conserved) U[0] = load_row<PL>(in + row_off(min(r0 + 3, p.n0 + 1)), p.plane_stride, jc8);
        if constexpr (PLM && A::shared_differences)
        {
            const State5 Dab = A::difference(Pa, Pb), Db0 = A::difference(Pb, P[0]);
            D[0] = A::difference(P[0], P[1]);
            const State5 Gb = A::plm_from_differences(Dab, Db0, lim);
            G[0] = A::plm_from_differences(Db0, D[0], lim);
            Fx[0] = A::template flux<RIEMANN, 0>(A::plus(Pb, Gb, lim), A::minus(P[0], G[0], lim), gl);
        }
        else if constexpr (PLM)
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

    // Error contract (status_device.hpp): an updated density that is not > 0 and a recovered pressure that is not >= 0 are recorded with
    // their kind (NaN apart) and flat cell index row * n1 + col. The tests are wave-wide votes - scalar branches never taken in a healthy run.
    StatusAcc acc;
    const uint32_t n1u = (uint32_t) p.n1, colu = (uint32_t) col;
    if (__any(!(P[0][4] >= 0.0) || !(P[1][4] >= 0.0)))      // the chunk's first two rows (the row loop checks rows r + 2)
    {
        if (writes && !(P[0][4] >= 0.0)) acc.note_value(P[0][4], MH_STATUS_NEG_PRESSURE, (uint32_t) r0 * n1u + colu);
        if (writes && !(P[1][4] >= 0.0) && r0 + 1 < p.n0) acc.note_value(P[1][4], MH_STATUS_NEG_PRESSURE, (uint32_t) (r0 + 1) * n1u + colu);
    }
    // second prefetch stage: row r+3 is in flight in Upre while row r is processed, row r+4 is issued at its top.
    // Two rows (10 loads of 512 B) in flight per wave keep ~40 KB outstanding per CU, enough to cover HBM latency
    // at this kernel's bandwidth (one row in flight left the first RK stage latency-bound).
    State5 Upre;
    if constexpr (! A::recompute_conserved) Upre = load_row<PL>(in + row_off(min(r0 + 3, p.n0 + 1)), p.plane_stride, jc8);

    // one row; K0 = ring slot of row r (compile-time), K1 / K2 = slots of rows r+1 / r+2
    auto row_step = [&] (int r, auto k0) __attribute__((always_inline))
    {
        constexpr int K0 = decltype(k0)::value, K1 = (K0 + 1) % 3, K2 = (K0 + 2) % 3;

        // issue the load of row r+4 (clamped to the stored ghost range; unused past the chunk end)
        const int rp = min(r + 4, p.n0 + 1);
        State5 Unext;
        if constexpr (A::recompute_conserved) U[K1] = load_row<PL>(in + row_off(rp), p.plane_stride, jc8);      // row r+1 was converted a row ago
        else                                  Unext = load_row<PL>(in + row_off(rp), p.plane_stride, jc8);
        State5 Ubase;
        if constexpr (COMBINE) Ubase = load_row<PL>(p.u_base + row_off(r), p.plane_stride, jc8);      // (requesting it a row earlier: no faster, measured)

        // ---- axis 0: flux through face r+1/2
        P[K2] = A::c2p(U[K2], gl);
        ring_put(K2, U[K2]);
        // a negative (or NaN) pressure: the strict arithmetic turns it into NaN sound speeds that reach the density check below, the fast
        // arithmetic's guarded inverse root would not - so it is flagged where it appears (once per cell and stage)
        const bool bad_pressure = !(P[K2][4] >= 0.0);
        if constexpr (PLM && A::shared_differences)
        {
            D[K1] = A::difference(P[K1], P[K2]);
            G[K1] = A::plm_from_differences(D[K0], D[K1], lim);
            Fx[K1] = A::template flux<RIEMANN, 0>(A::plus(P[K0], G[K0], lim), A::minus(P[K1], G[K1], lim), gl);
        }
        else if constexpr (PLM)
        {
            G[K1] = A::plm(P[K0], P[K1], P[K2], lim);
            Fx[K1] = A::template flux<RIEMANN, 0>(A::plus(P[K0], G[K0], lim), A::minus(P[K1], G[K1], lim), gl);
        }
        else
        {
            Fx[K1] = A::template flux<RIEMANN, 0>(P[K0], P[K1], gl);
        }

        // ---- axis 1: this lane computes the flux through its LEFT face (between lane-1 and lane)
        State5 Fy_lo, Fy_hi;
        if constexpr (PLM && A::shared_differences)
        {
            const State5 Dr = A::difference(P[K0], from_right_p<PL>(P[K0]));
            const State5 Gy = A::plm_from_differences(from_left_p<PL>(Dr), Dr, lim);
            const State5 SL = from_left_p<PL>(A::plus(P[K0], Gy, lim));
            Fy_lo = A::template flux<RIEMANN, 1>(SL, A::minus(P[K0], Gy, lim), gl);
        }
        else if constexpr (PLM)
        {
            const State5 Gy = A::plm(from_left_p<PL>(P[K0]), P[K0], from_right_p<PL>(P[K0]), lim);
            const State5 SL = from_left_p<PL>(A::plus(P[K0], Gy, lim));       // left neighbour's right-going face state
            Fy_lo = A::template flux<RIEMANN, 1>(SL, A::minus(P[K0], Gy, lim), gl);
        }
        else
        {
            Fy_lo = A::template flux<RIEMANN, 1>(from_left_p<PL>(P[K0]), P[K0], gl);
        }
        Fy_hi = from_right_p<PL>(Fy_lo);

        // ---- conservative update (+ RK combine)
        State5 Un, Uc;
        if constexpr (lds_ring)                    Uc = ring_get(K0);
        else if constexpr (A::recompute_conserved) Uc = A::p2c(P[K0], gl);
        else                                  Uc = U[K0];
#pragma unroll
        for (int q = 0; q < 5; ++q)
        {
            if (! live(q)) { Un[q] = 0.0; continue; }
            const double u1 = A::update2(Uc[q], Fx[K0][q], Fx[K1][q], Fy_lo[q], Fy_hi[q], p.cx, p.cy);
            if constexpr (COMBINE) Un[q] = A::combine(Ubase[q], u1, p.weight);
            else                   Un[q] = u1;
        }
        const bool bad_density = !(Un[0] > 0.0);          // catches <= 0 and NaN
        if (__any(bad_pressure || bad_density))
        {
            if (writes && bad_pressure && r + 2 < p.n0) acc.note_value(P[K2][4], MH_STATUS_NEG_PRESSURE, (uint32_t) (r + 2) * n1u + colu);
            if (writes && bad_density) acc.note_value(Un[0], MH_STATUS_NEG_DENSITY, (uint32_t) r * n1u + colu);
        }

        if (writes)
        {
            store_row(p.u_out + row_off(r), p.plane_stride, col8, Un);
            // keep the physical axis-0 ghost rows of the output current (edge rows only: wave-uniform, cold)
            if (r < HALO || r >= p.n0 - HALO)
            {
                if (p.bc_lo0 == 0 && r == 0) { store_row(p.u_out + row_off(-1), p.plane_stride, col8, Un); store_row(p.u_out + row_off(-2), p.plane_stride, col8, Un); }
                if (p.bc_hi0 == 1 && r < HALO) store_row(p.u_out + row_off(p.n0 + r), p.plane_stride, col8, Un);
                if (p.bc_hi0 == 0 && r == p.n0 - 1) { store_row(p.u_out + row_off(p.n0), p.plane_stride, col8, Un); store_row(p.u_out + row_off(p.n0 + 1), p.plane_stride, col8, Un); }
                if (p.bc_lo0 == 1 && r >= p.n0 - HALO) store_row(p.u_out + row_off(r - p.n0), p.plane_stride, col8, Un);
            }
        }
        if constexpr (! A::recompute_conserved)
        {
            U[K0] = Upre;       // slot of row r now holds row r+3 ...
            Upre = Unext;       // ... and row r+4 stays in flight
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

    acc.commit(p.status);
}

template<class A, int RIEMANN, bool PLM, bool COMBINE>
__global__ __launch_bounds__(WAVE * WAVES_PER_BLOCK, (COMBINE ? A::min_waves_per_simd : A::min_waves_first_stage))
void euler2d_stage_kernel(Stage2dParams p)
{
    stage_body<A, RIEMANN, PLM, COMBINE>(p, (int) blockIdx.x, (int) gridDim.x);
}

// The slab stepper orders its two streams with events. An event recorded by hipEventRecord is a separate marker packet behind
// the kernel; handed to the launch itself (hipExtLaunchKernel's stopEvent) it rides on the dispatch packet's own completion signal,
// one packet less on the chain between consecutive stages. With a start event too the pair brackets exactly the kernel (profiling
// without marker packets around the launch). Both arrive as launch parameters (LaunchEvents, launch.hpp).
template<class A, int RIEMANN, bool PLM, bool COMBINE>
static hipError_t launch(const Stage2dParams& p, hipStream_t stream, const LaunchEvents& ev)
{
    const int nwaves = p.nstrips * p.nchunks;
    const int nblocks = (nwaves + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
    if (ev.stop)
        hipExtLaunchKernelGGL((euler2d_stage_kernel<A, RIEMANN, PLM, COMBINE>), dim3(nblocks), dim3(WAVE * WAVES_PER_BLOCK), 0, stream, ev.start, ev.stop, 0, p);
    else
        hipLaunchKernelGGL((euler2d_stage_kernel<A, RIEMANN, PLM, COMBINE>), dim3(nblocks), dim3(WAVE * WAVES_PER_BLOCK), 0, stream, p);
    return hipGetLastError();
}

hipError_t euler2d_stage_launch(const mh_euler_cart_desc* d, const double* u_in, const double* u_base, double* u_out,
                                double dt, double weight, int row_begin, int row_end, int32_t* status, hipStream_t stream, LaunchEvents ev)
{
    return euler2d_stage_launch2(d, u_in, u_base, u_out, dt, weight, row_begin, row_end, 0, 0, status, stream, ev);
}

// fills the kernel parameters of a stage launch over [row_begin, row_end) (+ an optional second range); returns its workgroup count
static int build_params(Stage2dParams& p, const mh_euler_cart_desc* d, const double* u_in, const double* u_base, double* u_out,
                        double dt, double weight, int row_begin, int row_end, int row_begin2, int row_end2, int32_t* status)
{
    p.u_in = u_in;
    p.u_base = u_base;
    p.u_out = u_out;
    p.status = status;
    p.n0 = d->n[0];
    p.n1 = d->n[1];
    p.plane_stride = p.n1;
    p.row_stride = 5L * p.n1;
    p.row_begin = row_begin;
    p.row_end = row_end;
    p.nstrips = (p.n1 + STRIP - 1) / STRIP;
    if (d->chunk_rows > 0) p.chunk_rows = d->chunk_rows;
    else
    {
        // default. These kernels hold 2 waves per SIMD (190-226 VGPRs), i.e. 2048 resident waves on the 256 CUs.
        // A thin slab is fastest when the whole launch fits ONE residency round: chunk = rows / floor(2048 / strips)
        // (measured on 512/1024/2048 x 4096 slabs: 0.135 / 0.269 / 0.522 ms per step against 0.158 / 0.297 / 0.534 with
        // short chunks). Large grids need many rounds anyway: 32 rows per wave (7 % prologue overhead) is as fast
        // there and keeps halo re-reads close together in time (L2 hits).
        const long rows = (long) (row_end - row_begin) + (row_end2 - row_begin2);
        const long chunks_max = 2048 / p.nstrips > 0 ? 2048 / p.nstrips : 1;
        const long c = (rows + chunks_max - 1) / chunks_max;
        p.chunk_rows = (int) (c > 96 ? 32 : (c < 4 ? 4 : c));
    }
    p.chunk_rows2 = p.chunk_rows;
    // Graded tail (large single-range launches only). The workgroups of a launch are dispatched in index order and the last residency
    // round ends ragged: for about one wave duration the chip runs half empty (measured: the same kernels reach 59 / 67 % of the HBM
    // roofline at 16384^2 against 50 / 61 % at 4096^2). Giving the LAST rows to short waves shortens that window; their extra prologue
    // work is paid on a small fraction of the rows only. The descriptor's tail_rows / tail_chunk_rows override (tail_rows < 0 = off).
    // An explicit tail applies to any single-range launch with more rows than the tail (that is how the tests reach this path on
    // small grids); the default applies to large launches only.
    const bool tail_given = d->tail_rows != 0;
    if (row_end2 == row_begin2 && (tail_given || (d->chunk_rows == 0 && p.chunk_rows == 32)))
    {
        int tail_rows = 512, tail_chunk = 8;          // measured at 4096^2 (3 alternating runs): 0.752 -> 0.739 ms per step; 384,8 the same; 256,16 and 1024,16 no gain
        if (tail_given) { tail_rows = d->tail_rows; tail_chunk = d->tail_chunk_rows > 0 ? d->tail_chunk_rows : 8; }
        if (tail_rows > 0 && tail_chunk >= 2 && row_end - row_begin >= (tail_given ? tail_rows + 1 : 4 * tail_rows))
        {
            row_begin2 = row_end - tail_rows;
            row_end2 = row_end;
            row_end = row_begin2;
            p.row_end = row_end;
            p.chunk_rows2 = tail_chunk;
        }
    }
    p.nchunks_a = (row_end - row_begin + p.chunk_rows - 1) / p.chunk_rows;
    p.row_begin2 = row_begin2;
    p.row_end2 = row_end2;
    p.nchunks = p.nchunks_a + (row_end2 - row_begin2 + p.chunk_rows2 - 1) / p.chunk_rows2;
    p.tail_blocks_per_xcd = p.chunk_rows2 != p.chunk_rows ? (int) (((long) p.nstrips * (p.nchunks - p.nchunks_a) / WAVES_PER_BLOCK) >> 3) : 0;
    p.bc_lo0 = d->bc_lo0;
    p.bc_hi0 = d->bc_hi0;
    p.bc1 = d->bc_transverse;
    p.gamma = d->gamma;
    p.theta = d->plm_theta;
    p.cx = dt / d->dl[0];
    p.cy = dt / d->dl[1];
    p.weight = weight;
    if (p.nchunks <= 0) return 0;
    return (p.nstrips * p.nchunks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
}

// d->planar > 0 (a stepper that verified it at upload, or the caller's assertion): the planar instantiations, built for PLM
static int variant_key(const mh_euler_cart_desc* d, bool combine)
{
    const bool plm = d->plm_theta >= 0.0;
    return (d->planar > 0 && plm ? 16 : 0) | (d->arith == MH_ARITH_FAST ? 8 : 0) | (d->riemann == MH_RIEMANN_HLLC ? 4 : 0) | (plm ? 2 : 0) | (combine ? 1 : 0);
}

hipError_t euler2d_stage_launch2(const mh_euler_cart_desc* d, const double* u_in, const double* u_base, double* u_out,
                                 double dt, double weight, int row_begin, int row_end, int row_begin2, int row_end2,
                                 int32_t* status, hipStream_t stream, LaunchEvents ev)
{
    Stage2dParams p;
    const int nblocks_total = build_params(p, d, u_in, u_base, u_out, dt, weight, row_begin, row_end, row_begin2, row_end2, status);
    if (nblocks_total <= 0)
    {
        if (ev.stop)          // nothing to launch: the events still have to fire
        {
            if (ev.start) (void) hipEventRecord(ev.start, stream);
            return hipEventRecord(ev.stop, stream);
        }
        return hipSuccess;
    }

    switch (variant_key(d, weight != 1.0))
    {
        case 0:  return launch<StrictArith, 0, false, false>(p, stream, ev);
        case 1:  return launch<StrictArith, 0, false, true >(p, stream, ev);
        case 2:  return launch<StrictArith, 0, true,  false>(p, stream, ev);
        case 3:  return launch<StrictArith, 0, true,  true >(p, stream, ev);
        case 4:  return launch<StrictArith, 1, false, false>(p, stream, ev);
        case 5:  return launch<StrictArith, 1, false, true >(p, stream, ev);
        case 6:  return launch<StrictArith, 1, true,  false>(p, stream, ev);
        case 7:  return launch<StrictArith, 1, true,  true >(p, stream, ev);
        case 8:  return launch<FastArith, 0, false, false>(p, stream, ev);
        case 9:  return launch<FastArith, 0, false, true >(p, stream, ev);
        case 10: return launch<FastArith, 0, true,  false>(p, stream, ev);
        case 11: return launch<FastArith, 0, true,  true >(p, stream, ev);
        case 12: return launch<FastArith, 1, false, false>(p, stream, ev);
        case 13: return launch<FastArith, 1, false, true >(p, stream, ev);
        case 14: return launch<FastArith, 1, true,  false>(p, stream, ev);
        case 15: return launch<FastArith, 1, true,  true >(p, stream, ev);
        case 18: return launch<StrictArithPlanar, 0, true, false>(p, stream, ev);
        case 19: return launch<StrictArithPlanar, 0, true, true >(p, stream, ev);
        case 22: return launch<StrictArithPlanar, 1, true, false>(p, stream, ev);
        case 23: return launch<StrictArithPlanar, 1, true, true >(p, stream, ev);
        case 26: return launch<FastArithPlanar, 0, true, false>(p, stream, ev);
        case 27: return launch<FastArithPlanar, 0, true, true >(p, stream, ev);
        case 30: return launch<FastArithPlanar, 1, true, false>(p, stream, ev);
        case 31: return launch<FastArithPlanar, 1, true, true >(p, stream, ev);
    }
    return hipErrorInvalidValue;
}

// row-range guard (row_check.hpp): what this translation unit's kernels asked for; false in product builds
MH_ROW_RANGE_READER(rows_requested_euler2d)

} // namespace mh
