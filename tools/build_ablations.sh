#!/bin/bash
# tools/build_ablations.sh <source file> <macro> <bits> ...  — one library per COMPILE-time ablation of one kernel source:
#     tools/build_ablations.sh conv_wreg.hip WG_ABLATE_CT 4 5 36     ->  tools/lib/libglass_conv_wreg_4.so ...
#     tools/build_ablations.sh conv_d0.hip   D0_ABLATE_CT 1 2 128
#     tools/build_ablations.sh upfir.hip     U_ABLATE_CT  1 4 16
# = the release objects with that one object replaced (bit meanings: the macro's comment in the source).  Compare with tools/layer_ab.py.
# Why compile time: a run-time `if (abl & bit)` around MFMA groups made the developer build 20 % slower than the kernel it was measuring, and
# skipping an epilogue let hipcc delete the MFMAs of accumulators nobody read any more ("Round 6").
set -e
src=$1; macro=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root/clip_glass_amd/csrc"
make -j8 > /dev/null
base=${src%.hip}
mkdir -p ../../tools/lib /tmp/abl
for b in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value -Wno-unused-variable -Wno-unused-but-set-variable -D$macro=$b -c $src -o /tmp/abl/${base}_$b.o
  objs=$(ls build/*.o | grep -v "build/$base.o")
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/abl/${base}_$b.o -o ../../tools/lib/libglass_${base}_$b.so
  echo "built tools/lib/libglass_${base}_$b.so"
done
