#!/bin/bash
# Dev helper: libmara_hip.so with ONE source compiled with extra flags, every other object taken from the product build (mara3_amd/build).
# usage: scripts/build_variant_one.sh <name> <file.hip> "<extra flags>"  ->  build/variants/<name>/libmara_hip.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; extra=$3
out=$ROOT/build/variants/$name
mkdir -p $out/obj
cd $ROOT/mara3_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -Wno-unused-value $extra -c $src -o $out/obj/${src%.hip}.o
objs=""
for o in $ROOT/mara3_amd/build/*.o; do
  b=$(basename $o)
  if [ "$b" = "${src%.hip}.o" ]; then objs="$objs $out/obj/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libmara_hip.so $objs -ldl
echo $out/libmara_hip.so
