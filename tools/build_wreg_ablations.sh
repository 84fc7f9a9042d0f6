#!/bin/bash
# tools/build_wreg_ablations.sh <bits> ...  — one library per compile-time ablation of conv_wreg.hip (WG_ABLATE_CT): tools/lib/libglass_wg<bits>.so
# = the release objects with conv_wreg.o replaced.  Bits: 1 epilogue, 2 its global stores, 4 MFMAs, 8 patch fragment reads, 32 the ring's requests.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root/clip_glass_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tools/lib /tmp/wgabl
for b in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value -Wno-unused-variable -DWG_ABLATE_CT=$b -c conv_wreg.hip -o /tmp/wgabl/conv_wreg_$b.o
  objs=$(ls build/*.o | grep -v conv_wreg.o)
  hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/wgabl/conv_wreg_$b.o -o ../../tools/lib/libglass_wg$b.so
  echo "built tools/lib/libglass_wg$b.so"
done
