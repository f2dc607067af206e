#!/bin/bash
# Round-3 rocprofv3 evidence (one MI355X). usage: bash scripts/profile_r3.sh headline|headline2|configs
# Every rocprofv3 run is its own pass (kernel trace, or ONE group of PMC counters); the program goes directly after `--`.
export TMPDIR=/tmp
PART=${1:-headline}
OUT=$PWD/gpurun_out/prof_r3
mkdir -p $OUT
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
FLOP="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64"
pmc_passes () {   # tag, then the command
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_trace -- "$@" > $OUT/${tag}_trace.json 2> $OUT/${tag}_trace.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${tag}_fetch -- "$@" > $OUT/${tag}_fetch.json 2> $OUT/${tag}_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${tag}_write -- "$@" > $OUT/${tag}_write.json 2> $OUT/${tag}_write.err
  rocprofv3 --pmc $SQ1 --output-format csv -d $OUT/${tag}_sq -- "$@" > $OUT/${tag}_sq.json 2> $OUT/${tag}_sq.err
  rocprofv3 --pmc $FLOP --output-format csv -d $OUT/${tag}_flop -- "$@" > $OUT/${tag}_flop.json 2> $OUT/${tag}_flop.err
  rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/${tag}_grbm -- "$@" > $OUT/${tag}_grbm.json 2> $OUT/${tag}_grbm.err
  echo "done $tag"
}
if [ $PART = headline ]; then
  rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
  B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --single-arith --blocks 1"
  pmc_passes fast_hllc $B --arith fast --riemann hllc
  pmc_passes fast_hllc_two $B --arith fast --riemann hllc --no-fuse
  pmc_passes strict_hlle $B --arith strict --riemann hlle
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- python3 scripts/calib_fetch.py > $OUT/calib_f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- python3 scripts/calib_fetch.py > $OUT/calib_w.log 2>&1
elif [ $PART = headline_trace ]; then
  # kernel trace only, over a run long enough for the cold launches (the first ~25 after an idle period run up to twice as long) not to
  # carry the average: 200 timed steps instead of 20
  B="python3 bench.py --steps 200 --warmup 3 --no-cpu-baseline --single-arith --blocks 1"
  for v in "fast_hllc --arith fast --riemann hllc" "fast_hllc_two --arith fast --riemann hllc --no-fuse" "strict_hlle --arith strict --riemann hlle" "fast_hlle --arith fast --riemann hlle" "strict_hllc --arith strict --riemann hllc"; do
    set -- $v; tag=$1; shift
    rm -rf $OUT/${tag}_trace
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_trace -- $B "$@" > $OUT/${tag}_trace.json 2> $OUT/${tag}_trace.err
    echo "done $tag"
  done
elif [ $PART = headline2 ]; then
  B="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --single-arith --blocks 1"
  pmc_passes fast_hlle $B --arith fast --riemann hlle
  pmc_passes strict_hllc $B --arith strict --riemann hllc
else
  pmc_passes c3 python3 bench_configs.py --config c3 --steps 5 --no-cpu-baseline
  pmc_passes c5 python3 bench_configs.py --config c5 --steps 3 --grid 384 --no-cpu-baseline
  EXE=$PWD/mara3_amd/host/mara_hip
  ( cd $OUT && pmc_passes c4 $EXE cloud nr=4096 num_decades=1 rk_order=2 reconstruct_method=2 plm_theta=1.2 max_steps=6 cpi=0 arith=fast outdir=x; rm -rf x )
  ( cd $OUT && pmc_passes c4s $EXE cloud nr=4096 num_decades=1 rk_order=2 reconstruct_method=2 plm_theta=1.2 max_steps=4 cpi=0 arith=strict outdir=x; rm -rf x )
fi
python3 scripts/pmc_summary.py $OUT > $OUT/summary_$PART.txt 2>&1
for d in $OUT/*_trace; do echo "== $d"; find $d -name "*kernel_stats.csv" | head -1 | xargs head -5; done > $OUT/kernel_stats_$PART.txt
find $OUT -name "*.csv" -size +2M -delete
find $OUT -name "*.db" -delete
tail -n 3 $OUT/*.err | grep -v "amdgpu.ids" | tail -20
