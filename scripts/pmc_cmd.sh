#!/bin/bash
# Kernel-trace stats + SQ counter passes of an arbitrary python command (dev helper).
# usage: bash scripts/pmc_cmd.sh <tag> <script.py> [args...]     (the program goes directly after `--`, no wrappers)
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/t.log 2> $OUT/t.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq1 -- python3 "$@" > $OUT/b1.log 2> $OUT/e1.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -- python3 "$@" > $OUT/b2.log 2> $OUT/e2.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 "$@" > $OUT/b3.log 2> $OUT/e3.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 "$@" > $OUT/b4.log 2> $OUT/e4.err
cat $OUT/t.log
for f in $OUT/*.err; do tail -n 1 $f; done
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs cat | head -8
python3 scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1; cat $OUT/summary.txt
