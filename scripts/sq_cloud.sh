#!/bin/bash
# SQ counters + kernel trace of the cloud stage kernels (dev helper). usage: bash scripts/sq_cloud.sh <tag> [arith]
TAG=${1:-cloud}; ARITH=${2:-fast}; MORE=$3          # e.g. "fuse=-1" (two launches per RK2 step) or "fuse=1 chunk_rows=64"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
EXE=$PWD/mara3_amd/host/mara_hip
ARGS="cloud nr=4096 num_decades=1 rk_order=2 reconstruct_method=2 plm_theta=1.2 max_steps=6 cpi=0 arith=$ARITH outdir=x $MORE"
cd $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $EXE $ARGS > t.log 2> t.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/sq1 -- $EXE $ARGS > b1.log 2> e1.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/sq2 -- $EXE $ARGS > b2.log 2> e2.err
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/grbm -- $EXE $ARGS > b3.log 2> e3.err
rm -rf x
cd - > /dev/null
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs head -4
python3 scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +1M -delete
