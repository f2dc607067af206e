#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the stage kernels of configs 3-5 (dev helper).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_cfg
mkdir -p $OUT
for c in c3 c5; do
  extra=""; [ $c = c5 ] && extra="--steps 3 --grid 384"
  [ $c = c3 ] && extra="--steps 5"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${c}_fetch -- python3 bench_configs.py --config $c --no-cpu-baseline $extra > $OUT/${c}_f.json 2> $OUT/${c}_f.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${c}_write -- python3 bench_configs.py --config $c --no-cpu-baseline $extra > $OUT/${c}_w.json 2> $OUT/${c}_w.err
done
( cd $OUT && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c4_fetch -- $OLDPWD/mara3_amd/host/mara_hip cloud nr=4096 num_decades=1 rk_order=2 plm_theta=1.2 max_steps=4 cpi=0 arith=fast outdir=x > c4_f.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/c4_write -- $OLDPWD/mara3_amd/host/mara_hip cloud nr=4096 num_decades=1 rk_order=2 plm_theta=1.2 max_steps=4 cpi=0 arith=fast outdir=x > c4_w.log 2>&1; rm -rf x )
python3 scripts/pmc_summary.py $OUT 2>&1 | grep -E "==|stage_kernel|update_kernel|flux_kernel" | grep -v duration > $OUT/summary.txt
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +1M -delete
