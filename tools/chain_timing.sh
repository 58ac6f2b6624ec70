#!/bin/bash
# where the cycles of a prediction-chain wave go in the few-pictures cuts: chain.hip built with -DHM_CHAIN_TIMING (s_memtime laps
# per phase, summed over the waves of a launch into its sync region), printed by hm_batch_check with HM_CHAIN_TIMING_PRINT=1
mkdir -p gpurun_out
cd heif-decoder-lib_amd/csrc
rm -f build/hip_chain.o
make HIPFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I. -I/opt/rocm/include -DHM_CHAIN_TIMING" >/dev/null 2>&1
cd ../..
{
HM_CHAIN_TIMING_PRINT=1 timeout 600 python3 tools/few_pictures_probe.py 2>&1 | grep "k_chain phases" | sort | uniq -c | sort -rn | head -12
echo "== one 12 MP grid through the plugin"
HM_CHAIN_TIMING_PRINT=1 timeout 300 python3 tools/plugin_probe.py 2>&1 | grep "k_chain phases" | tail -3
} > gpurun_out/chain_timing.log 2>&1
cd heif-decoder-lib_amd/csrc; rm -f build/hip_chain.o; make >/dev/null 2>&1
