#!/bin/bash
# A/B builds of chain.hip in the few-pictures regime (tools/few_pictures_probe.py): VARIANTS as in tools/probe_chain.sh
mkdir -p gpurun_out
cd heif-decoder-lib_amd/csrc
IFS='|' read -ra VS <<< "${VARIANTS:--DHM_NONE=1}"
{
for v in "${VS[@]}"; do
  rm -f build/hip_chain.o
  make HIPFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I. -I/opt/rocm/include $v" >/dev/null 2>&1
  echo -n "variant [$v]: "
  (cd ../.. && python3 tools/few_pictures_probe.py 2>/dev/null | tail -1)
done
} > ../../gpurun_out/probe_few.log 2>&1
rm -f build/hip_chain.o; make >/dev/null 2>&1
