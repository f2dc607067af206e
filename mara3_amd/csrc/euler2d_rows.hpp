// Row access and lane-to-lane movement shared by the 2-D Euler stage kernels (euler2d.hip: one stage per launch; euler2d_fused.hip: both
// RK2 stages in one launch).
#pragma once
#include <hip/hip_runtime.h>
#include "euler_device.hpp"

namespace mh {

// value of the lane on the left (lane-1) / right (lane+1); the edge lane reads 0 (its result is never used)
__device__ inline double from_left(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);   // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double from_right(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);   // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ inline State5 from_left(const State5& s)
{
    State5 r;
#pragma unroll
    for (int q = 0; q < 5; ++q) r[q] = from_left(s[q]);
    return r;
}
__device__ inline State5 from_right(const State5& s)
{
    State5 r;
#pragma unroll
    for (int q = 0; q < 5; ++q) r[q] = from_right(s[q]);
    return r;
}

// PLANAR: the fourth variable (third momentum / velocity) is identically zero and does not travel
template<bool PLANAR>
__device__ inline State5 from_left_p(const State5& s)
{
    State5 r;
#pragma unroll
    for (int q = 0; q < 5; ++q) r[q] = (PLANAR && q == 3) ? 0.0 : from_left(s[q]);
    return r;
}
template<bool PLANAR>
__device__ inline State5 from_right_p(const State5& s)
{
    State5 r;
#pragma unroll
    for (int q = 0; q < 5; ++q) r[q] = (PLANAR && q == 3) ? 0.0 : from_right(s[q]);
    return r;
}

// One row of one field through a buffer resource: the row pointer is wave-uniform (scalar registers), the lane
// contributes a 32-bit byte offset and the variable a scalar offset, so a row costs five buffer instructions and
// no vector address arithmetic (cdna_hip_programming.md T8). The descriptor covers exactly the row block
// (5 variables x n1 doubles): anything outside returns 0 / is dropped by the hardware range check.
using b64_t = decltype(__builtin_amdgcn_raw_buffer_load_b64(__amdgpu_buffer_rsrc_t(), 0, 0, 0));

// PLANAR: the fourth variable (third momentum) is identically zero in the field and is not read
template<bool PLANAR = false>
__device__ inline State5 load_row(const double* row, long plane_stride, unsigned lane_bytes)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(row), 0, (int) (5 * plane_stride * 8), 0x00020000);
    State5 U;
#pragma unroll
    for (int q = 0; q < 5; ++q)
    {
        if (PLANAR && q == 3) { U[q] = 0.0; continue; }
        U[q] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, lane_bytes, (unsigned) (q * plane_stride * 8), 0));
    }
    return U;
}
__device__ inline void store_row(double* row, long plane_stride, unsigned lane_bytes, const State5& U)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(row, 0, (int) (5 * plane_stride * 8), 0x00020000);
#pragma unroll
    for (int q = 0; q < 5; ++q)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(b64_t, U[q]), rs, lane_bytes, (unsigned) (q * plane_stride * 8), 0);
}

} // namespace mh
