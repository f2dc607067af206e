// The stage kernel of the `binary` path (design notes and citations: binary.hip) as a header, so that the STRICT instantiations
// (binary.hip: reference operation order) and the FAST ones (binary_fast.hip: the scheme's glue - face states, viscous stress, source
// terms, totals, update - with gathered factors and explicit FMAs, like the FAST leaf functions) come from one source. No file is compiled
// with FMA contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <type_traits>
#include "launch.hpp"
#include "binary_device.hpp"
#include "status_device.hpp"
#include "row_check.hpp"

namespace mh {

static constexpr int BWAVE = 64;
static constexpr int BHALO = 2;
static constexpr int BSTRIP = BWAVE - 2 * BHALO;
static constexpr int BWAVES_PER_BLOCK = 4;
static constexpr int NPART = 8;    // per-wave partial sums: torque[2], fx[2], fy[2], mass_ejected, L_ejected
static constexpr int NBLK = 10;    // per-block results: mass_acc[2], L_acc[2], px_acc[2], py_acc[2], work[2]

struct BinaryStageParams
{
    const double* u_in;
    const double* u_base;
    double*       u_out;
    const double* u_init;
    const double* br;
    const double* xv;
    const double* yv;
    double*       partials;   // [nwaves][NPART]
    int32_t*      status;
    int32_t*      status_clear;   // or null: two status words of ANOTHER block, zeroed by this launch for the stage issued behind it (binary_api.hip: eager stage)
    const double* xvg;        // x vertices of the WHOLE mesh (xv = xvg + row0: this band's)
    int    n, chunk_rows, nstrips, nchunks;
    // the rows this launch covers: chunks [0, seg0_chunks) cut rows [seg0_begin, seg0_end), the others [seg1_begin, seg1_end) (one launch
    // for both edges of a band, binary.hip); wave_base: where this launch's waves stand in the partial sums of the stage
    int    seg0_begin, seg0_end, seg0_chunks, seg1_begin, seg1_end, wave_base;
    int    n0, row0, ext0;    // band of the mesh held by this field: rows [row0, row0 + n0); ext0: its ghost rows belong to other bands
    double theta, dt, weight;
    BinaryConsts c;
};

__device__ inline double bdpp_left(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);   // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double bdpp_right(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);   // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline State3 bdpp_left(const State3& s) { State3 r; for (int q = 0; q < 3; ++q) r[q] = bdpp_left(s[q]); return r; }
__device__ inline State3 bdpp_right(const State3& s) { State3 r; for (int q = 0; q < 3; ++q) r[q] = bdpp_right(s[q]); return r; }

using bb64_t = decltype(__builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t(), 0, 0, 0));

// one row of a 3-plane field: wave-uniform row pointer (scalar), per-lane byte offset
__device__ inline State3 load_row3(const double* row, int n, unsigned lane_bytes)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(row), 0, 3 * n * 8, 0x00020000);
    State3 U;
#pragma unroll
    for (int q = 0; q < 3; ++q)
        U[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, lane_bytes, (unsigned) (q * n * 8), 0));
    return U;
}
__device__ inline void store_row3(double* row, int n, unsigned lane_bytes, const State3& U)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(row, 0, 3 * n * 8, 0x00020000);
#pragma unroll
    for (int q = 0; q < 3; ++q)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(bb64_t, U[q]), rs, lane_bytes, (unsigned) (q * n * 8), 0);
}
__device__ inline double load_row1(const double* row, int n, unsigned lane_bytes)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(row), 0, n * 8, 0x00020000);
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, lane_bytes, 0, 0));
}

__device__ inline double wave_sum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

#ifndef MH_BIN_WAVES_PER_SIMD
#define MH_BIN_WAVES_PER_SIMD 2
#endif
template<class A, bool COMBINE, bool QFORM>
__global__ __launch_bounds__(BWAVE * BWAVES_PER_BLOCK, MH_BIN_WAVES_PER_SIMD)
void binary_stage_kernel(BinaryStageParams p)
{
    const int nblocks = gridDim.x;
    int b = blockIdx.x;
    {
        const int per_xcd = nblocks >> 3;
        if (b < per_xcd * 8) b = (b & 7) * per_xcd + (b >> 3);
    }
    const int w = __builtin_amdgcn_readfirstlane(b * BWAVES_PER_BLOCK + (int) (threadIdx.x >> 6));
    if (w >= p.nstrips * p.nchunks) return;
    const int lane = threadIdx.x & 63;
    const int chunk_of_launch = w / p.nstrips;
    const int strip = w - chunk_of_launch * p.nstrips;
    const int n = p.n, n0 = p.n0;
    const bool second = chunk_of_launch >= p.seg0_chunks;
    const int chunk = second ? chunk_of_launch - p.seg0_chunks : chunk_of_launch;
    const int r0 = (second ? p.seg1_begin : p.seg0_begin) + chunk * p.chunk_rows;
    const int r1 = min(r0 + p.chunk_rows, second ? p.seg1_end : p.seg0_end);

    // column of this lane: `col` is the un-wrapped index (positions), `jc` the periodic image (data)
    const int col = strip * BSTRIP - BHALO + lane;
    int jc = col < 0 ? col + n : (col >= n ? col - n : col);
    jc = min(max(jc, 0), n - 1);
    const bool writes = lane >= BHALO && lane < BWAVE - BHALO && col < n;
    const unsigned jc8 = (unsigned) jc * 8u, col8 = (unsigned) (writes ? col : 0) * 8u;

    // positions: the LEFT face of this lane sits at the un-wrapped vertex (so the two sides of the periodic seam differ);
    // everything cell-centred belongs to the cell whose data the lane holds, i.e. the wrapped column jc
    const double yv_lo = p.yv[min(max(col, 0), n)];
    const double yc = (p.yv[jc] + p.yv[jc + 1]) * 0.5;
    const double dy = p.yv[jc + 1] - p.yv[jc];
    // centre of the cell that the data of (possibly ghost) row r belongs to: the periodic image within the whole mesh
    auto xc_of = [&p, n] (int r) { const int g = p.row0 + r; const int rw = g < 0 ? g + n : (g >= n ? g - n : g); return (p.xvg[rw] + p.xvg[rw + 1]) * 0.5; };

    const BinaryConsts& c = p.c;
    const typename A::Ctx k = A::make(c);
    const double theta = p.theta, dt = p.dt;
    const long row_stride = 3L * n;
    const int rows_hi = n0 + 1;                        // the rows that exist: -2 .. n0 + 1 (row_check.hpp)
    auto row_off = [row_stride, rows_hi] (int r) { (void) rows_hi; return (long) (MH_ROW(r, -BHALO, rows_hi) + BHALO) * row_stride; };
    const double* in = p.u_in;

    // ring slots (index = row mod 3 relative to the chunk start), as in euler2d.hip:
    //   U[k], P[k]: rows r, r+1, r+2     Gx[k], Gy[k]: slopes of rows r, r+1     Fx[k]: faces r, r+1 (times dy)
    // Measured and NOT taken (MH_BIN_LDS_RING=1, profiles/r03/ab_binary.jsonl): as in euler2d.hip, the conserved rows waiting for their
    // update in a per-wave LDS ring, U[] holding only the rows that are loaded and not yet converted (no register-to-register moves, 239
    // instead of 246 VGPRs): 0.117 against 0.1145 ms per stage at 2048^2 on one box - the kernel is not short of registers in a way that
    // seven of them would mend, and the six LDS instructions per row are not free.
#ifndef MH_BIN_LDS_RING
#define MH_BIN_LDS_RING 0
#endif
    constexpr bool lds_ring = A::arith == MH_ARITH_FAST && MH_BIN_LDS_RING;
    __shared__ double own_rows[lds_ring ? BWAVES_PER_BLOCK : 1][3][3][BWAVE];
    const int wave_in_block = (int) (threadIdx.x >> 6);
    auto ring_put = [&] (int slot, const State3& raw)
    {
        if constexpr (lds_ring)
        {
#pragma unroll
            for (int q = 0; q < 3; ++q) own_rows[wave_in_block][slot][q][lane] = raw[q];
        }
    };
    auto ring_get = [&] (int slot) -> State3
    {
        State3 Uq;
#pragma unroll
        for (int q = 0; q < 3; ++q) Uq[q] = own_rows[lds_ring ? wave_in_block : 0][slot][q][lane];
        return Uq;
    };
    State3 U[3], P[3], Gx[3], Gy[3], Fx[3];
    {
        const State3 Pa = A::template c2p<QFORM>(load_row3(in + row_off(r0 - 2), n, jc8), xc_of(r0 - 2), yc);
        const State3 Pb = A::template c2p<QFORM>(load_row3(in + row_off(r0 - 1), n, jc8), xc_of(r0 - 1), yc);
        U[0] = load_row3(in + row_off(r0), n, jc8);
        U[1] = load_row3(in + row_off(r0 + 1), n, jc8);
        U[2] = load_row3(in + row_off(r0 + 2), n, jc8);
        P[0] = A::template c2p<QFORM>(U[0], xc_of(r0), yc);
        P[1] = A::template c2p<QFORM>(U[1], xc_of(r0 + 1), yc);
        ring_put(0, U[0]);
        ring_put(1, U[1]);
        if constexpr (lds_ring) U[0] = load_row3(in + row_off(min(r0 + 3, n0 + 1)), n, jc8);      // slot of row r0 + 3 (rows r0, r0 + 1 are converted)
        const State3 Gxb = A::plm_per_length(Pa, Pb, P[0], theta, k);
        const State3 Gyb = A::plm_per_length(bdpp_left(Pb), Pb, bdpp_right(Pb), theta, k);
        Gx[0] = A::plm_per_length(Pb, P[0], P[1], theta, k);
        Gy[0] = A::plm_per_length(bdpp_left(P[0]), P[0], bdpp_right(P[0]), theta, k);
        Fx[0] = binary_face_flux<A, 0, QFORM>(c, k, p.xv[r0], yc, Pb, P[0], Gxb, Gx[0], Gyb, Gy[0]);
#pragma unroll
        for (int q = 0; q < 3; ++q) Fx[0][q] = A::carried(Fx[0][q] * dy);
    }
    State3 Upre;
    if constexpr (! lds_ring) Upre = load_row3(in + row_off(min(r0 + 3, n0 + 1)), n, jc8);

    double part[NPART];
#pragma unroll
    for (int k = 0; k < NPART; ++k) part[k] = 0.0;
    StatusAcc acc;        // validate_u (scheme.cpp:726-752) as status bits + first failing cell r * n + col (status_device.hpp)
    double sigma_new = 0.0;

    // x vertices of the row in flight, carried from row to row: one scalar load per row, requested a row before its first use (loading
    // xv[r], xv[r + 1] at the top of row r parked the wave on the scalar cache's latency every row: SQ_WAIT_ANY 40 % of the wave cycles)
    double xv_lo = p.xv[r0], xv_hi = p.xv[r0 + 1];
    auto row_step = [&] (int r, auto k0) __attribute__((always_inline))
    {
        constexpr int K0 = decltype(k0)::value, K1 = (K0 + 1) % 3, K2 = (K0 + 2) % 3;
        const double xv_next = p.xv[min(r + 2, n0)];
        State3 Unext;
        if constexpr (lds_ring) U[K1] = load_row3(in + row_off(min(r + 4, n0 + 1)), n, jc8);       // row r + 1 was converted a row ago
        else                    Unext = load_row3(in + row_off(min(r + 4, n0 + 1)), n, jc8);
        const State3 Uinit = load_row3(p.u_init + row_off(r), n, jc8);
        const double brate = load_row1(p.br + (long) r * n, n, jc8);
        State3 Ubase;
        if constexpr (COMBINE) Ubase = load_row3(p.u_base + row_off(r), n, jc8);

        const double xlo = xv_lo, xhi = xv_hi;
        const double xc = (xlo + xhi) * 0.5;
        const double dx = xhi - xlo;

        // ---- axis 0: slopes of row r+1, flux through face r+1 (at x = xv[r+1])
        P[K2] = A::template c2p<QFORM>(U[K2], xc_of(r + 2), yc);
        ring_put(K2, U[K2]);
        Gx[K1] = A::plm_per_length(P[K0], P[K1], P[K2], theta, k);
        Gy[K1] = A::plm_per_length(bdpp_left(P[K1]), P[K1], bdpp_right(P[K1]), theta, k);
        Fx[K1] = binary_face_flux<A, 0, QFORM>(c, k, xhi, yc, P[K0], P[K1], Gx[K0], Gx[K1], Gy[K0], Gy[K1]);
#pragma unroll
        for (int q = 0; q < 3; ++q) Fx[K1][q] = A::carried(Fx[K1][q] * dy);

        // ---- axis 1: this lane's LEFT face (at y = yv[col]), handed to the left neighbour as its right face
        State3 Fy_lo = binary_face_flux<A, 1, QFORM>(c, k, xc, yv_lo, bdpp_left(P[K0]), P[K0], bdpp_left(Gy[K0]), Gy[K0], bdpp_left(Gx[K0]), Gx[K0]);
#pragma unroll
        for (int q = 0; q < 3; ++q) Fy_lo[q] = Fy_lo[q] * dx;
        const State3 Fy_hi = bdpp_right(Fy_lo);

        State3 Un;
        if constexpr (A::arith == MH_ARITH_FAST && ! QFORM)
        {
            // ---- MH_ARITH_FAST: the source terms of :345-411 and the update of :568-587 with their common factors gathered - dt, the
            // two bodies' sink rates and the floor term (all multiply u0), 1 / dA = 1 / h^2 on this uniform mesh - and written as FMAs:
            // the same formulas within the mode's tolerance (1e-12 of the field scale), a third of the instructions of the generic form below
            State3 u0;
            if constexpr (lds_ring) u0 = ring_get(K0);
            else                    u0 = U[K0];
            const double dA = dx * dy, dtA = dt * dA;
            double fg[2][2], rate = 0.0;
#pragma unroll
            for (int bdy = 0; bdy < 2; ++bdy)
            {
                const double d0 = xc - c.body[5 * bdy + 1], d1 = yc - c.body[5 * bdy + 2];
                A::gravity(c, bdy, d0, d1, u0[0], fg[bdy]);
                rate += binary_sink_rate<A>(c, d0, d1);
            }
            const double w0 = __builtin_fma(-rate, dt, u0[0] < c.floor_sigma ? 1e-2 : 0.0);       // sinks and floor
            const double bw = brate * dt;
            double sb[3], s[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) sb[q] = (Uinit[q] - u0[q]) * bw;
            s[0] = __builtin_fma(u0[0], w0, sb[0]);
            s[1] = __builtin_fma(u0[1], w0, __builtin_fma(fg[0][0] + fg[1][0], dt, sb[1]));
            s[2] = __builtin_fma(u0[2], w0, __builtin_fma(fg[0][1] + fg[1][1], dt, sb[2]));
            if (writes)
            {
#pragma unroll
                for (int bdy = 0; bdy < 2; ++bdy)
                {
                    part[0 + bdy] = __builtin_fma(__builtin_fma(xc, fg[bdy][1], -yc * fg[bdy][0]), dtA, part[0 + bdy]);
                    part[2 + bdy] = __builtin_fma(fg[bdy][0], dtA, part[2 + bdy]);
                    part[4 + bdy] = __builtin_fma(fg[bdy][1], dtA, part[4 + bdy]);
                }
                part[6] = __builtin_fma(sb[0], dA, part[6]);
                part[7] = __builtin_fma(__builtin_fma(xc, sb[2], -yc * sb[1]), dA, part[7]);
            }
            const double rA = dt * k.inv_h2;
#pragma unroll
            for (int q = 0; q < 3; ++q)
            {
                const double l = (Fx[K1][q] - Fx[K0][q]) + (Fy_hi[q] - Fy_lo[q]);
                const double u1 = __builtin_fma(-l, rA, u0[q] + s[q]);
                if constexpr (COMBINE) Un[q] = __builtin_fma(u1, p.weight, Ubase[q] * (1.0 - p.weight));
                else                   Un[q] = u1;
                if (q == 0) sigma_new = u1;
            }
        }
        else
        {
        // ---- source terms :345-411
        State3 u0;
        if constexpr (lds_ring) u0 = ring_get(K0);
        else                    u0 = U[K0];
        const double dA = dx * dy;
        double fg[2][2], s_grav[2][3], s_sink[2][3], s_buffer[3], s_floor[3];
#pragma unroll
        for (int bdy = 0; bdy < 2; ++bdy)
        {
            const double d0 = xc - c.body[5 * bdy + 1], d1 = yc - c.body[5 * bdy + 2];
            A::gravity(c, bdy, d0, d1, u0[0], fg[bdy]);
            s_grav[bdy][0] = 0.0 * dt;
            s_grav[bdy][1] = fg[bdy][0] * dt;
            s_grav[bdy][2] = fg[bdy][1] * dt;
            const double rate = binary_sink_rate<A>(c, d0, d1);
#pragma unroll
            for (int q = 0; q < 3; ++q) s_sink[bdy][q] = -u0[q] * rate * dt;
        }
        const double fl = u0[0] < c.floor_sigma ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < 3; ++q)
        {
            s_buffer[q] = (Uinit[q] - u0[q]) * brate * dt;
            s_floor[q] = u0[q] * 1e-2 * fl;
        }
        if constexpr (QFORM)
        {
            // source_terms_q :417-466: gravity as (s_r, l_z) sources, and the geometrical source of the s_r equation
            // (source_terms_conserved_angmom physics_iso2d.hpp:277-285, ramped down within gst_suppr_radius of the origin)
            // in the place of the density-floor term
#pragma unroll
            for (int bdy = 0; bdy < 2; ++bdy)
            {
                s_grav[bdy][1] = (xc * fg[bdy][0] + yc * fg[bdy][1]) * dt;
                s_grav[bdy][2] = (xc * fg[bdy][1] - yc * fg[bdy][0]) * dt;
            }
            const double a = -(xc * xc + yc * yc) / c.sr2;
            double e = 0.0;
            if (__any(a > -750.0)) e = exp(a);            // exp underflows to exactly 0 below -745.2
            const double ramp = 1.0 - e;
            const State3& pc = P[K0];
            const double Ek = 0.5 * pc[0] * (pc[1] * pc[1] + pc[2] * pc[2]);
            const double pg = pc[0] * A::cs2(c, k, xc, yc);
            s_floor[0] = 0.0 * ramp * dt;
            s_floor[1] = (Ek + pg) * 2.0 * ramp * dt;
            s_floor[2] = 0.0 * ramp * dt;
        }
        if (writes)
        {
#pragma unroll
            for (int bdy = 0; bdy < 2; ++bdy)
            {
                part[0 + bdy] = part[0 + bdy] + (QFORM ? s_grav[bdy][2] : (xc * s_grav[bdy][2] - yc * s_grav[bdy][1])) * dA;
                part[2 + bdy] = part[2 + bdy] + fg[bdy][0] * dt * dA;
                part[4 + bdy] = part[4 + bdy] + fg[bdy][1] * dt * dA;
            }
            part[6] = part[6] + s_buffer[0] * dA;
            part[7] = part[7] + (QFORM ? s_buffer[2] : (xc * s_buffer[2] - yc * s_buffer[1])) * dA;
        }

        // ---- update :568-587 (+ RK combine)
        {
            double l[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) l[q] = ((Fx[K1][q] - Fx[K0][q]) + (Fy_hi[q] - Fy_lo[q])) * dt;
            A::over_area(l, dA);
#pragma unroll
            for (int q = 0; q < 3; ++q)
            {
                const double s = s_grav[0][q] + s_grav[1][q] + s_sink[0][q] + s_sink[1][q] + s_buffer[q] + s_floor[q];
                const double u1 = u0[q] - l[q] + s;
                if constexpr (COMBINE) Un[q] = Ubase[q] * (1.0 - p.weight) + u1 * p.weight;
                else                   Un[q] = u1;
                if (q == 0) sigma_new = u1;
            }
        }
        }
        if (__any(!(sigma_new >= 0.0)))          // validate_u :726-752 (and NaN); a scalar branch never taken in a healthy run
        {
            if (writes && !(sigma_new >= 0.0)) acc.note_value(sigma_new, MH_STATUS_NEG_DENSITY, (uint32_t) (p.row0 + r) * (uint32_t) n + (uint32_t) col);
        }
        if (writes)
        {
            store_row3(p.u_out + row_off(r), n, col8, Un);
            if (! p.ext0)          // the whole mesh in this field: periodic ghost rows of the output (a band's come from its neighbours)
            {
                if (r < BHALO) store_row3(p.u_out + row_off(n + r), n, col8, Un);
                if (r >= n - BHALO) store_row3(p.u_out + row_off(r - n), n, col8, Un);
            }
        }
        if constexpr (! lds_ring)
        {
            U[K0] = Upre;
            Upre = Unext;
        }
        xv_lo = xv_hi;
        xv_hi = xv_next;
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

#pragma unroll
    for (int k = 0; k < NPART; ++k)
    {
        const double s = wave_sum(part[k]);
        if (lane == 0) p.partials[(long) (p.wave_base + w) * NPART + k] = s;
    }
    acc.commit(p.status);
    if (p.status_clear && w == 0 && lane == 0) { p.status_clear[0] = 0; p.status_clear[1] = 0; }
}

// launches the instantiation (COMBINE, QFORM) of arithmetic A; the FAST ones live in binary_fast.hip
// done (or null): an event that fires when THIS launch completes. It rides on the dispatch packet's own completion signal (hipExtLaunchKernel,
// as the 2-D slab stepper's launches do) instead of being recorded behind the launch, where it is a marker packet between two stage kernels of
// the main stream. Measured at 2048^2 (round 5, profiles/r05/ab_c3_event_on_launch.txt; MH_BIN_EVENT_ON_LAUNCH=0 records it as before): FAST
// 21.3 - 21.5 Gzones/s either way (the 6.7 us between stage kernels in a rocprofv3 trace are the tracer's), STRICT 11.27 against 11.05.
template<class A>
inline hipError_t binary_stage_dispatch(const BinaryStageParams& p, dim3 grid, dim3 block, hipStream_t stream, bool combine, bool qform, hipEvent_t done = nullptr)
{
#define MH_BIN_LAUNCH(C, Q) do { if (done) hipExtLaunchKernelGGL((binary_stage_kernel<A, C, Q>), grid, block, 0, stream, nullptr, done, 0, p); \
                                 else      hipLaunchKernelGGL((binary_stage_kernel<A, C, Q>), grid, block, 0, stream, p); } while (0)
    switch ((qform ? 2 : 0) | (combine ? 1 : 0))
    {
        case 0: MH_BIN_LAUNCH(false, false); break;
        case 1: MH_BIN_LAUNCH(true,  false); break;
        case 2: MH_BIN_LAUNCH(false, true ); break;
        case 3: MH_BIN_LAUNCH(true,  true ); break;
    }
#undef MH_BIN_LAUNCH
    return hipGetLastError();
}
hipError_t binary_stage_dispatch_fast(const BinaryStageParams& p, dim3 grid, dim3 block, hipStream_t stream, bool combine, bool qform, hipEvent_t done = nullptr);

} // namespace mh
