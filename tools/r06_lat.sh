#!/bin/bash
# r06: where the cycles of a wave go in the few-pictures regime (ONE 12 MP image = 48 tiles, the launcher's one-chain cut): instrumented builds of chain.hip
# HM_CHAIN_TIMING = 1 cycles per phase, 2 events per phase, 3 cycles of the luma waves with the service phase split into work and waiting
FL="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I. -I/opt/rocm/include"
for t in ${TIMINGS:-1 2 3}; do
  (cd heif-decoder-lib_amd/csrc && rm -f build/hip_chain.o && make HIPFLAGS="$FL -DHM_CHAIN_TIMING=$t" >/dev/null 2>&1)
  for n in ${IMAGES:-1 8}; do
    echo "== HM_CHAIN_TIMING=$t, $n image(s)"
    HM_CHAIN_TIMING_PRINT=1 timeout 600 python3 bench.py --quick --no-parity --steps 3 --images $n 2>&1 | grep -E "k_chain phases" | tail -1
  done
done
(cd heif-decoder-lib_amd/csrc && rm -f build/hip_chain.o && make >/dev/null 2>&1)
