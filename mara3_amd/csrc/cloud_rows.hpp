// Parameters, geometry records, row access and lane-to-lane movement shared by the `cloud` stage kernels (cloud.hip: one stage per
// launch; cloud_fused.hip: both RK2 stages in one launch). Reference lines: src/subprog_cloud.cpp:260-290 (geometry), :511-584 (advance).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "euler_device.hpp"
#include "srhd_device.hpp"

namespace mh {

struct CloudParams
{
    const double* u_in;
    const double* u_base;
    double*       u_out;
    const double* rv;          // radial vertices of the GLOBAL grid (device)
    const double* dmu;         // [nq]   -cos q_{j+1} - -cos q_j
    const double* sinq;        // [nq+1] sin q_j
    const double* cotq;        // [nq]   tan(pi/2 - theta_c)
    const double* rowf;        // [nr_global][8] per-row factors    (MH_ARITH_FAST; mh_cloud_pack_geometry)
    const double* colf;        // [nq][8]        per-column factors
    const double* inflow;      // [5][nq] primitives of the inner ghost row
    int32_t*      status;
    long   plane_stride, row_stride;
    int    n0, n1;             // local radial rows, polar columns
    int    row_offset;         // global index of local row 0
    int    row_begin, row_end, chunk_rows, nstrips, nchunks;
    int    row_begin2, row_end2, chunk_rows2, nchunks_a, tail_blocks_per_xcd;   // graded tail: chunks >= nchunks_a are short and march [row_begin2, row_end2)
    int    bc_lo0, bc_hi0;     // MH_BC_INFLOW / MH_BC_OUTFLOW (physical) or MH_BC_EXTERNAL (slab cut)
    double gamma, theta, tfloor, dt, weight;
};

__device__ inline double dpp_left(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double dpp_right(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline State5 dpp_left(const State5& s) { State5 r; for (int q = 0; q < 5; ++q) r[q] = dpp_left(s[q]); return r; }
__device__ inline State5 dpp_right(const State5& s) { State5 r; for (int q = 0; q < 5; ++q) r[q] = dpp_right(s[q]); return r; }
__device__ inline State5 times_zero(const State5& s) { State5 r; for (int q = 0; q < 5; ++q) r[q] = s[q] * 0.0; return r; }
// planar states (component 3 is +0.0 in every lane: nothing to move, (+0) * 0.0 = +0)
template<bool PLANAR> __device__ inline State5 dpp_left_p(const State5& s) { State5 r; for (int q = 0; q < 5; ++q) r[q] = PLANAR && q == 3 ? 0.0 : dpp_left(s[q]); return r; }
template<bool PLANAR> __device__ inline State5 dpp_right_p(const State5& s) { State5 r; for (int q = 0; q < 5; ++q) r[q] = PLANAR && q == 3 ? 0.0 : dpp_right(s[q]); return r; }
template<bool PLANAR> __device__ inline State5 times_zero_p(const State5& s) { State5 r; for (int q = 0; q < 5; ++q) r[q] = PLANAR && q == 3 ? 0.0 : s[q] * 0.0; return r; }

// wave-uniform radial geometry of global row i
struct RowGeom
{
    double rr_lo, rr_hi;     // r_i r_i, r_{i+1} r_{i+1}   ((r + r) * 0.5 == r exactly)
    double d3;               // r_{i+1}^3 - r_i^3          (its axis-1 midpoint (x + x) * 0.5 == x exactly)
    double rcdr;             // ((r_i + r_{i+1}) * 0.5) * (r_{i+1} - r_i)
    double rc;               // (r_i + r_{i+1}) * 0.5
};
__device__ inline RowGeom row_geometry(const double* rv, int i)
{
    const double r0 = rv[i], r1 = rv[i + 1];
    RowGeom g;
    g.rr_lo = r0 * r0;
    g.rr_hi = r1 * r1;
    g.d3 = r1 * r1 * r1 - r0 * r0 * r0;
    g.rc = (r0 + r1) * 0.5;
    g.rcdr = g.rc * (r1 - r0);
    return g;
}

struct ColGeom { double dmu, sin_lo, sin_hi, cot; };

// everything one cell's update needs from the grid, per row of the march
struct CellGeom { double dv, inv_dv, nAr_lo, nAr_hi, nAq_lo, nAq_hi, rc, inv_rc; };

// the five variables of one stored row through a buffer resource: wave-uniform row pointer (scalar registers), per-lane byte offset,
// scalar plane offset - five buffer instructions and no vector address arithmetic (as euler2d.hip)
using cb64_t = decltype(__builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t(), 0, 0, 0));
// (PLANAR: the azimuthal plane is known to hold +0.0 - not read; the store still writes it, the output buffer's plane is not known to)
template<bool PLANAR = false>
__device__ inline State5 cloud_load_row(const double* row, long plane, unsigned lane_bytes)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(row), 0, (int) (5 * plane * 8), 0x00020000);
    State5 U;
#pragma unroll
    for (int q = 0; q < 5; ++q)
    {
        if (PLANAR && q == 3) U[q] = 0.0;
        else U[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, lane_bytes, (unsigned) (q * plane * 8), 0));
    }
    return U;
}
__device__ inline void cloud_store_row(double* row, long plane, unsigned lane_bytes, const State5& U)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(row, 0, (int) (5 * plane * 8), 0x00020000);
#pragma unroll
    for (int q = 0; q < 5; ++q) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(cb64_t, U[q]), rs, lane_bytes, (unsigned) (q * plane * 8), 0);
}

__device__ inline double cell_volume(const RowGeom& rg, const ColGeom& cg, const Recip& three)
{
    return divide(rg.d3 * cg.dmu * 2 * M_PI, three);
}

} // namespace mh
