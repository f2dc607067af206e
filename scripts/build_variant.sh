#!/bin/bash
# Dev helper: build libmara_hip.so with extra compiler flags into build/variants/<name>/libmara_hip.so (for A/B runs through
# MARA_HIP_LIBRARY). usage: scripts/build_variant.sh <name> "<extra flags>"
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; extra=$2
out=$ROOT/build/variants/$name
mkdir -p $out/obj
cd $ROOT/mara3_amd/csrc
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -Wno-unused-value $extra"
objs=""
for f in *.hip; do
  per_file=""; [ $f = euler3d_fast.hip ] && per_file="-mllvm -amdgpu-sched-strategy=max-ilp"      # as the Makefile
  /opt/rocm/bin/hipcc $HIPFLAGS $per_file -c $f -o $out/obj/${f%.hip}.o &
  objs="$objs $out/obj/${f%.hip}.o"
done
for f in twobody binary_host; do
  g++ -std=c++17 -O2 -fPIC -ffp-contract=off -fno-fast-math -Wall -c $f.cpp -o $out/obj/$f.o &
  objs="$objs $out/obj/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libmara_hip.so $objs -ldl
echo $out/libmara_hip.so
