#!/bin/bash
# which cut (a wave per picture / pair of rows / row / chain) for how many 512x512 tiles
mkdir -p gpurun_out
{
for n in 48 192 768 1536 3072; do
 for m in 0 1 2 3; do
  echo -n "tiles $n HM_CHAIN_PAIRS=$m: "; HM_CLASS_TILES=$n HM_CHAIN_PAIRS=$m timeout 600 python3 tools/bench_classes.py 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print({k: v['k_recon_ms'] for k, v in d.items() if k in ('8bit_420_ctb32','8bit_420_ctb64')})"
 done
done
} > gpurun_out/r03_thresh.log 2>&1
