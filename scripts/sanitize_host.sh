#!/bin/bash
# AddressSanitizer + UBSan over the HOST-side C++ of the product on the CPU box (GPU sanitizers are not available on this pool):
#   * csrc/twobody.cpp, csrc/binary_host.cpp rebuilt instrumented and linked with the product's other objects -> the CPU tests that call them
#     through the C ABI (tests/test_binary_host_cpu.py, test_tree_curve_order_cpu.py, test_two_body_cpu.py, test_abi_cpu.py)
#   * host/h5_selftest, host/h5_tool (the HDF5 writer / reader): their own round trip and tests/test_h5_format_cpu.py
# usage: bash scripts/sanitize_host.sh     (needs `make -C mara3_amd/csrc` done; writes under /tmp/mara_san only)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
S=/tmp/mara_san; mkdir -p $S
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g"
cd $ROOT/mara3_amd/csrc
for f in twobody binary_host; do g++ -std=c++17 -fPIC $SAN -ffp-contract=off -fno-fast-math -c $f.cpp -o $S/$f.o; done
objs=""
for o in ../build/*.o; do b=$(basename $o); case $b in twobody.o|binary_host.o) objs="$objs $S/$b";; *) objs="$objs $o";; esac; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o $S/libmara_hip.so $objs -ldl
cd $ROOT/mara3_amd/host
for t in h5_selftest h5_tool; do g++ -std=c++17 $SAN -ffp-contract=off -I../../include -idirafter ${HDF5_INC:-/opt/conda/include} $t.cpp -o $S/$t -ldl; done
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
( cd $S && ./h5_selftest $S/roundtrip.h5 )
cd $ROOT
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) MARA_HIP_LIBRARY=$S/libmara_hip.so \
  python -m pytest tests/test_binary_host_cpu.py tests/test_tree_curve_order_cpu.py tests/test_two_body_cpu.py tests/test_abi_cpu.py -x -q
cp mara3_amd/host/h5_tool $S/h5_tool.product && cp $S/h5_tool mara3_amd/host/h5_tool
python -m pytest tests/test_h5_format_cpu.py -x -q; rc=$?
cp $S/h5_tool.product mara3_amd/host/h5_tool
exit $rc
