#!/bin/bash
# the 16-bit classes at full load in the finer cuts of a picture (their LDS per wave limits a wave per picture to 4-13 waves per CU)
mkdir -p gpurun_out
{
for e in "HM_CHAIN_PAIRS=0" "HM_CHAIN_PAIRS=1" "HM_CHAIN_PAIRS=2" "HM_CHAIN_PAIRS=3"; do
  echo "== $e"; env $e HM_CHAIN_DEBUG=1 HM_QUAD_CLASS=1 HM_CLASS_ONLY=12bit_422_ctb64,10bit_422_ctb32,10bit_420_ctb32,8bit_420_ctb64 HM_CLASS_TILES=${TILES:-18432} timeout 900 python3 tools/bench_classes.py 2>&1 | grep -v amdgpu.ids | awk '!seen[$0]++' | tr -d '\n' | sed 's/\[k_chain\]/\n  [k_chain]/g; s/{ *"/\n  {"/; s/}, /},\n   /g'; echo
done
} > gpurun_out/r03_cuts16.log 2>&1
