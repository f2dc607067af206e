// Device side of the error contract of include/mara_hip.h: a stage cannot throw, so where the reference
// throws (src/physics_srhd.hpp:430-449, src/subprog_binary_scheme.cpp:726-752) or where an Euler state becomes unphysical
// the kernels record WHAT went wrong (mh_status bits) and WHERE it went wrong first (the smallest flat cell index of the
// launch's field) in a two-word device block:
//     status[0]   OR of mh_status bits
//     status[1]   max over failing cells of (0xFFFFFFFF - flat_index), as uint32; 0 = nothing recorded
// Both words start from 0 (a plain memset) and are order-independent (atomicOr / atomicMax), so the result does not depend on
// the launch geometry. The checks themselves stay out of the row loops' instruction stream: a kernel tests its conditions
// with one wave-wide vote (a scalar branch that is never taken in a healthy run) and calls note() only behind it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mara_hip.h"

namespace mh {

struct StatusAcc
{
    uint32_t bits = 0;
    uint32_t first = 0;          // 0xFFFFFFFF - smallest failing flat index seen by this lane (0: none)

    __device__ inline void note(uint32_t what, uint32_t flat_index)
    {
        bits |= what;
        const uint32_t key = 0xFFFFFFFFu - flat_index;
        first = key > first ? key : first;
    }
    // x is the offending value: NaN is reported as MH_STATUS_NAN, anything else as `otherwise`
    __device__ inline void note_value(double x, uint32_t otherwise, uint32_t flat_index)
    {
        note(x != x ? (uint32_t) MH_STATUS_NAN : otherwise, flat_index);
    }
    // wave-level reduction + one pair of atomics per wave that has something to say
    __device__ inline void commit(int32_t* status) const
    {
        if (! status) return;
        if (! __any(bits != 0)) return;
        uint32_t b = bits, f = first;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
        {
            b |= (uint32_t) __shfl_xor((int) b, off, 64);
            const uint32_t g = (uint32_t) __shfl_xor((int) f, off, 64);
            f = g > f ? g : f;
        }
        if ((threadIdx.x & 63) == 0)
        {
            atomicOr(status, (int32_t) b);
            atomicMax(reinterpret_cast<unsigned int*>(status) + 1, f);
        }
    }
};

} // namespace mh
