// The FAST instantiations of the `binary` stage kernel (binary_kernel.hpp; design notes in binary.hip) as their own translation unit,
// compiled with -ffp-contract=fast (Makefile). MH_ARITH_FAST promises the reference's formulas to 1e-12 of the field scale, not its
// operation order: the leaf functions of BinFast are written with explicit FMAs already, this lets the kernel's glue - face states
// (p +- g h / 2), viscous stress, the six source terms, the eight partial sums, the update and the RK average - contract too. Measured
// at 2048^2 (profiles/r03/ab_binary_contract.jsonl): the stage kernel's executed VALU instructions and its time.
#include "binary_kernel.hpp"

namespace mh {

hipError_t binary_stage_dispatch_fast(const BinaryStageParams& p, dim3 grid, dim3 block, hipStream_t stream, bool combine, bool qform)
{
    return binary_stage_dispatch<BinFast>(p, grid, block, stream, combine, qform);
}

} // namespace mh
