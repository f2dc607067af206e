// The FAST instantiations of the 3-D Euler stage kernel (euler3d_kernel.hpp; design notes in euler3d.hip) as their own translation unit:
// the Makefile compiles it with -mllvm -amdgpu-sched-strategy=max-ilp, which is worth 4 % on these kernels (3.74 -> 3.59 ms per 384^3 RK2
// step, profiles/r02/ab_scheduler_strategies_3d.jsonl) and costs the STRICT ones 1 %.
#include "euler3d_kernel.hpp"

namespace mh {

hipError_t euler3d_launch_fast(int key, const Stage3dParams& p, int nblocks, hipStream_t stream)
{
    switch (key)
    {
        case 0: return launch3<FastArith, 0, false, false>(p, nblocks, stream);
        case 1: return launch3<FastArith, 0, false, true >(p, nblocks, stream);
        case 2: return launch3<FastArith, 0, true,  false>(p, nblocks, stream);
        case 3: return launch3<FastArith, 0, true,  true >(p, nblocks, stream);
        case 4: return launch3<FastArith, 1, false, false>(p, nblocks, stream);
        case 5: return launch3<FastArith, 1, false, true >(p, nblocks, stream);
        case 6: return launch3<FastArith, 1, true,  false>(p, nblocks, stream);
        case 7: return launch3<FastArith, 1, true,  true >(p, nblocks, stream);
    }
    return hipErrorInvalidValue;
}

// row-range guard (row_check.hpp): what this translation unit's kernels asked for; false in product builds
MH_ROW_RANGE_READER(rows_requested_euler3d_fast)

} // namespace mh
