#!/bin/bash
# Dev measurement: kernel trace of the default `mara_hip binary` run (graded tree, 64 blocks of 24^2) and its own kzps without the profiler.
# usage (GPU box): bash scripts/profile_tree.sh <tag>
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-tree}
OUT=$ROOT/gpurun_out/prof_$TAG
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- $ROOT/mara3_amd/host/mara_hip binary max_iterations=400 steps_per_call=10 cpi=0 dfi=0 tsi=0 outdir=/tmp/bt_$TAG > $OUT.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0])):
    print("%-72s %5s calls  %9.1f ns" % (r["Name"][:72], r["Calls"], float(r["AverageNs"])))
PY
$ROOT/mara3_amd/host/mara_hip binary max_iterations=2000 steps_per_call=50 cpi=0 dfi=0 tsi=0 outdir=/tmp/bt2_$TAG | grep kzps | tail -4
