#!/bin/bash
# rocprofv3 passes for the round's profile evidence. Run on the GPU box from the repo root:
#   bash scripts/profile_r1.sh <tag>
# kernel-trace/stats and each PMC counter are collected in SEPARATE runs (gfx950: FETCH_SIZE and WRITE_SIZE
# do not fit in one pass; never combine --pmc with trace domains other than kernel-trace).
set -e
TAG=${1:-r01}; shift; ARGS="$@"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
cat /sys/fs/cgroup/cpu.max > $OUT/cpu_max.txt 2>/dev/null || true
nproc > $OUT/nproc.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --single-arith $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --single-arith $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --single-arith $ARGS > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- python3 scripts/calib_fetch.py > $OUT/calib_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- python3 scripts/calib_fetch.py > $OUT/calib_write.log 2>&1
