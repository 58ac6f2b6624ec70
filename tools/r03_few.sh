#!/bin/bash
mkdir -p gpurun_out
cd heif-decoder-lib_amd/csrc
IFS='|' read -ra VS <<< "$V"
for v in "${VS[@]}"; do
  rm -f build/hip_chain.o
  make HIPFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I. -I/opt/rocm/include $v" >/dev/null 2>&1
  echo "variant [$v]"
  (cd ../.. && timeout 600 python3 tools/few_pictures_probe.py 2>&1 | tail -1 | cut -c1-400)
done > ../../gpurun_out/r03_few.log 2>&1
rm -f build/hip_chain.o; make >/dev/null 2>&1
