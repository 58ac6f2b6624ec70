#!/bin/bash
# 8-bit CTB 64 at full load with A/B builds of chain.hip (VARIANTS as in tools/probe_chain.sh)
mkdir -p gpurun_out
cd heif-decoder-lib_amd/csrc
IFS='|' read -ra VS <<< "${VARIANTS:--DHM_NONE=1}"
{
for v in "${VS[@]}"; do
  rm -f build/hip_chain.o
  make HIPFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I. -I/opt/rocm/include $v" >/dev/null 2>&1
  echo -n "variant [$v]: "
  (cd ../.. && HM_CHAIN_DEBUG=1 HM_CLASS_ONLY=8bit_420_ctb64,8bit_420_ctb32 HM_CLASS_TILES=${TILES:-18432} timeout 900 python3 tools/bench_classes.py 2>&1 | grep -v amdgpu.ids | awk '!seen[$0]++' | tr -d '\n' | sed 's/\[k_chain\]/\n  [k_chain]/g; s/{ *"8bit/\n  {"8bit/'; echo)
done
} > ../../gpurun_out/r03_ctb64.log 2>&1
rm -f build/hip_chain.o; make >/dev/null 2>&1
