#!/bin/bash
# rocprofv3 kernel-trace summaries of the non-headline configs (dev helper; outputs under gpurun_out/prof_cfg).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_cfg
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -- python3 bench_configs.py --config c3 --no-cpu-baseline > $OUT/c3.json 2> $OUT/c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -- python3 bench_configs.py --config c5 --steps 5 --no-cpu-baseline > $OUT/c5.json 2> $OUT/c5.err
cd $OUT && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -- $OLDPWD/mara3_amd/host/mara_hip cloud nr=4096 num_decades=1 rk_order=2 reconstruct_method=2 plm_theta=1.2 max_steps=8 profile=1 outdir=c4out > $OUT/c4.log 2> $OUT/c4.err
cd $OLDPWD
for c in c3 c4 c5; do echo "== $c"; find $OUT/$c -name "*kernel_stats.csv" | head -1 | xargs head -6; done
tail -2 $OUT/c4.log
