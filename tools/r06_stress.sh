#!/bin/bash
# r06: tools/stress_cuts.py on the code as built, then its negative control - the early CTU start forced in the short non-alternating rings, where the launcher keeps the old rule
echo "== as built"; timeout 1500 python3 tools/stress_cuts.py ${REPS:-200} 2>&1 | grep -v amdgpu.ids | awk '/differ from the wave-per-picture cut: [1-9]/ || /TOTAL/'
FL="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I. -I/opt/rocm/include"
(cd heif-decoder-lib_amd/csrc && rm -f build/hip_chain.o && make HIPFLAGS="$FL -DHM_CHAIN_EARLY_FORCE" >/dev/null 2>&1)
echo "== negative control: -DHM_CHAIN_EARLY_FORCE"; timeout 1500 python3 tools/stress_cuts.py ${REPS:-200} 2>&1 | grep -v amdgpu.ids | awk '/differ from the wave-per-picture cut: [1-9]/ || /TOTAL/'
(cd heif-decoder-lib_amd/csrc && rm -f build/hip_chain.o && make >/dev/null 2>&1)
