#!/bin/bash
# Round-5 rocprofv3 evidence (one MI355X). usage: bash scripts/profile_r5.sh headline|headline2|configs|configs2
# Every rocprofv3 run is its own pass (kernel trace, or ONE group of PMC counters); the program goes directly after `--`.
# Kernel TRACES carry their own averages (verdict r4 #5): the headline variants over `bench.py --steps 1000` (about 1100 launches per kernel: the
# first ~25 launches after an idle period run up to twice as long and weigh 1 % there), C3 over 300 steps, C4 over 200, C5 over 110 (>= 100
# launches per kernel) - and the SAME process's JSON line is kept beside each trace (bench_under_rocprof_trace_<run>.json), so that
# tests/test_profiles_cpu.py can hold sum(min x launches per step) <= ms_per_step <= sum(avg x launches per step) x 1.05 for every run.
# The counter passes run short (20 steps / 5 / 6 / 3).
export TMPDIR=/tmp
PART=${1:-headline}
OUT=$PWD/gpurun_out/prof_r5
mkdir -p $OUT
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
FLOP="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"
counter_passes () {   # tag, then the command
  local tag=$1; shift
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${tag}_fetch -- "$@" > $OUT/${tag}_fetch.json 2> $OUT/${tag}_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${tag}_write -- "$@" > $OUT/${tag}_write.json 2> $OUT/${tag}_write.err
  rocprofv3 --pmc $SQ1 --output-format csv -d $OUT/${tag}_sq -- "$@" > $OUT/${tag}_sq.json 2> $OUT/${tag}_sq.err
  rocprofv3 --pmc $FLOP --output-format csv -d $OUT/${tag}_flop -- "$@" > $OUT/${tag}_flop.json 2> $OUT/${tag}_flop.err
  echo "done $tag counters"
}
trace_pass () {       # tag, then the command
  local tag=$1; shift
  rm -rf $OUT/${tag}_trace
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_trace -- "$@" > $OUT/${tag}_trace.json 2> $OUT/${tag}_trace.err
  echo "done $tag trace"
}
headline_variant () { # tag, bench flags
  local tag=$1; shift
  trace_pass $tag python3 bench.py --steps 1000 --warmup 3 --no-cpu-baseline --single-arith --blocks 1 "$@"
  counter_passes $tag python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --single-arith --blocks 1 "$@"
}
EXE=$PWD/mara3_amd/host/mara_hip
C4="cloud nr=4096 num_decades=1 rk_order=2 reconstruct_method=2 plm_theta=1.2 cpi=0 outdir=x"
if [ $PART = headline ]; then
  headline_variant fast_hllc --arith fast --riemann hllc
  headline_variant fast_hllc_general --arith fast --riemann hllc --no-planar
  headline_variant strict_hlle --arith strict --riemann hlle
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- python3 scripts/calib_fetch.py > $OUT/calib_f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- python3 scripts/calib_fetch.py > $OUT/calib_w.log 2>&1
elif [ $PART = headline2 ]; then
  headline_variant fast_hllc_two --arith fast --riemann hllc --no-fuse
  headline_variant fast_hlle --arith fast --riemann hlle
  headline_variant strict_hllc --arith strict --riemann hllc
elif [ $PART = configs ]; then
  trace_pass c3 python3 bench_configs.py --config c3 --steps 300 --warmup 60 --no-cpu-baseline
  counter_passes c3 python3 bench_configs.py --config c3 --steps 5 --no-cpu-baseline
  trace_pass c5 python3 bench_configs.py --config c5 --steps 110 --warmup 6 --no-cpu-baseline
  counter_passes c5 python3 bench_configs.py --config c5 --steps 3 --no-cpu-baseline
else
  ( cd $OUT && trace_pass c4 $EXE $C4 max_steps=200 arith=fast profile=1; counter_passes c4 $EXE $C4 max_steps=6 arith=fast; rm -rf x )
  ( cd $OUT && trace_pass c4two $EXE $C4 max_steps=200 arith=fast fuse=-1 profile=1; counter_passes c4two $EXE $C4 max_steps=6 arith=fast fuse=-1; rm -rf x )
  ( cd $OUT && trace_pass c4s $EXE $C4 max_steps=100 arith=strict profile=1; counter_passes c4s $EXE $C4 max_steps=4 arith=strict; rm -rf x )
fi
python3 scripts/pmc_summary.py $OUT > $OUT/summary_$PART.txt 2>&1
for d in $OUT/*_trace; do echo "== $d"; find $d -name "*kernel_stats.csv" | head -1 | xargs head -5; done > $OUT/kernel_stats_$PART.txt
# keep the small kernel_stats CSVs (they are copied into profiles/r05), drop the per-dispatch dumps
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -size +2M -delete
find $OUT -name "*.db" -delete
tail -n 3 $OUT/*.err | grep -v "amdgpu.ids" | tail -20
