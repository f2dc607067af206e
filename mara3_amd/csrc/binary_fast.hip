// The FAST instantiations of the `binary` stage kernel (binary_kernel.hpp; design notes in binary.hip) as their own translation unit (compile
// time). MH_ARITH_FAST promises the reference's formulas to 1e-12 of the field scale, not its operation order: the leaf functions of BinFast
// and the kernel's glue - face states (p +- g h / 2), viscous stress, the source terms, the partial sums, the update and the RK average - are
// written with explicit FMAs. Round 3 also let the compiler contract what was left (-ffp-contract=fast on this file): 1 % of the loop's
// instructions, no measurable time (profiles/r04/ab_c3_contract.jsonl) - dropped in round 4, so that the FAST bits do not depend on the
// compiler's contraction choices (advisor finding, round 3).
#include "binary_kernel.hpp"

namespace mh {

hipError_t binary_stage_dispatch_fast(const BinaryStageParams& p, dim3 grid, dim3 block, hipStream_t stream, bool combine, bool qform, hipEvent_t done)
{
    // the default disk (two-body sound speed, alpha viscosity, no cut-off) has its own instantiation: its row loop carries neither the branches
    // nor the code of the other forms (BinFastT<DISK>, binary_device.hpp) - same formulas, same bits
    if (BinFastDisk::serves(p.c)) return binary_stage_dispatch<BinFastDisk>(p, grid, block, stream, combine, qform, done);
    return binary_stage_dispatch<BinFast>(p, grid, block, stream, combine, qform, done);
}

// row-range guard (row_check.hpp): what this translation unit's kernels asked for; false in product builds
MH_ROW_RANGE_READER(rows_requested_binary_fast)

} // namespace mh
