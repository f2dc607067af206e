// Row-range guard of the row-marching kernels (build-time switch -DMH_CHECK_ROWS; `make -C mara3_amd/csrc check`).
//
// Every kernel that marches along axis 0 forms its row (plane) addresses from a row index through ONE function, and its software pipeline
// requests rows a few beyond the last one it uses. The rows that exist are the field's rows plus its stored ghost rows: [-2, n0 + 1]
// ([-4, n0 + 3] on the cut sides of the fused 2-D step, whose caller allocates four). A request outside them reads or writes outside the
// allocation; whether that faults depends on where the allocation happens to end (round 3: gpurun_out/r3o/one.log). Under MH_CHECK_ROWS
// every such index passes through checked_row(), which
//   * records the smallest and the largest index REQUESTED in two device words of its translation unit, and
//   * returns the index held to the rows that exist - a check build never leaves the allocation; it reports which kernel would have.
// tests/test_gpu_row_range.py runs each kernel family over the chunk / tail / segment combinations of the suite with the check library
// and asserts the recorded range; mh_debug_row_range (include/mara_hip.h) reads the words. Product builds: MH_ROW is the identity.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef MH_CHECK_ROWS
#include <limits.h>
namespace mh {
static __device__ int g_rows_requested[2] = {INT_MAX, INT_MIN};
__device__ inline int checked_row(int r, int lo, int hi)
{
    atomicMin(&g_rows_requested[0], r);
    atomicMax(&g_rows_requested[1], r);
    return r < lo ? lo : (r > hi ? hi : r);
}
// host side, one per translation unit: {smallest, largest} index requested since the last reset
static inline bool rows_requested_read(int32_t out[2], int reset)
{
    int h[2] = {INT_MAX, INT_MIN};
    if (hipDeviceSynchronize() != hipSuccess) return false;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rows_requested), sizeof h) != hipSuccess) return false;
    out[0] = h[0]; out[1] = h[1];
    if (reset)
    {
        const int fresh[2] = {INT_MAX, INT_MIN};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_rows_requested), fresh, sizeof fresh) != hipSuccess) return false;
    }
    return true;
}
} // namespace mh
#define MH_ROW(r, lo, hi) mh::checked_row((r), (lo), (hi))
#define MH_ROW_RANGE_READER(name) bool name(int32_t out[2], int reset) { return rows_requested_read(out, reset); }
#else
#define MH_ROW(r, lo, hi) (r)
#define MH_ROW_RANGE_READER(name) bool name(int32_t*, int) { return false; }
#endif
