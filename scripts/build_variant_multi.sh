#!/bin/bash
# Dev helper: libmara_hip.so with SEVERAL sources compiled with extra flags, every other object taken from the product build.
# usage: scripts/build_variant_multi.sh <name> "<extra flags>" file1.hip [file2.hip ...]  ->  build/variants/<name>/libmara_hip.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; extra=$2; shift 2
out=$ROOT/build/variants/$name
mkdir -p $out/obj
cd $ROOT/mara3_amd/csrc
repl=""
for src in "$@"; do
  fl=""
  [ $src = euler3d_fast.hip ] && fl="-mllvm -amdgpu-sched-strategy=max-ilp"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -Wno-unused-value $fl $extra -c $src -o $out/obj/${src%.hip}.o &
  repl="$repl ${src%.hip}.o"
done
wait
objs=""
for o in $ROOT/mara3_amd/build/*.o; do
  b=$(basename $o)
  case " $repl " in *" $b "*) objs="$objs $out/obj/$b";; *) objs="$objs $o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libmara_hip.so $objs -ldl
echo $out/libmara_hip.so
