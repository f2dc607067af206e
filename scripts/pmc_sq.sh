#!/bin/bash
# SQ-counter passes for the stage kernels (dev helper). usage: bash scripts/pmc_sq.sh <tag> [bench args]
TAG=${1:-sq}; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --single-arith "$@" > $OUT/b1.json 2> $OUT/e1.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/sq2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --single-arith "$@" > $OUT/b2.json 2> $OUT/e2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/grbm -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --single-arith "$@" > $OUT/b3.json 2> $OUT/e3.err
for f in $OUT/*.err; do tail -n 2 $f; done
