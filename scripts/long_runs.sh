#!/bin/bash
# End-to-end runs of the compiled hosts at the sub-programs' default parameters (dev helper; prints the tail of each log).
set -o pipefail
B=$PWD/mara3_amd/host/mara_hip
OUT=$PWD/gpurun_out/long; mkdir -p $OUT; cd $OUT
( time $B sedov outdir=sedov > sedov.log 2>&1 ); echo "sedov rc=$?"; tail -3 sedov.log
( time $B sedov newtonian=1 outdir=sedov_n > sedov_n.log 2>&1 ); echo "sedov newtonian rc=$?"; tail -2 sedov_n.log
( time $B cloud outdir=cloud > cloud.log 2>&1 ); echo "cloud rc=$?"; tail -3 cloud.log
( time $B cloud rk_order=2 arith=fast outdir=cloud_f > cloud_f.log 2>&1 ); echo "cloud fast rc=$?"; tail -2 cloud_f.log
( time $B binary depth=5 block_size=64 tfinal=0.25 steps_per_call=100 outdir=binary > binary.log 2>&1 ); echo "binary rc=$?"; grep -c "negative density" binary.log; tail -3 binary.log
( time $B binary depth=5 block_size=64 tfinal=0.25 steps_per_call=100 arith=fast conserve_linear_p=0 outdir=binary_q > binary_q.log 2>&1 ); echo "binary q fast rc=$?"; grep -c "negative density" binary_q.log; tail -2 binary_q.log
rm -rf sedov sedov_n cloud cloud_f binary binary_q
