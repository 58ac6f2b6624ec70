#!/bin/bash
# class / shape sweeps: every class through the split-chain kernels (HM_QUAD_CLASS=1) against the parser's own choice
mkdir -p gpurun_out
{
for q in 1 0; do
  echo "== shapes HM_QUAD_CLASS=$q"; HM_QUAD_CLASS=$q timeout 600 python3 tools/shape_probe.py 2>/dev/null | tail -1
done
for n in 1536 18432; do
 for q in 1 0; do
  echo "== classes $n tiles HM_QUAD_CLASS=$q"; HM_CLASS_TILES=$n HM_QUAD_CLASS=$q timeout 900 python3 tools/bench_classes.py 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print({k: v['k_recon_ms'] for k, v in d.items()})"
 done
done
} > gpurun_out/r03_sweeps.log 2>&1
