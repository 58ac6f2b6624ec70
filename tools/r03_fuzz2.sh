#!/bin/bash
# differential fuzzing of the end-of-round changes: the pre-pass in runs of CTUs (forced), in the default cut and per chain
mkdir -p gpurun_out
{
for e in "HM_RESID_SEGS=3" "HM_RESID_SEGS=16 HM_CHAIN_PAIRS=3" "HM_RESID_SEGS=5 HM_CHAIN_SHARE=2"; do
  for seed in 311 312; do echo "== fuzz_gpu seed $seed, all classes on the split-chain path, $e"; env HM_QUAD_CLASS=1 $e timeout 900 python3 tools/fuzz_gpu.py $seed 2>&1 | grep -v amdgpu.ids | tail -1; done
done
echo "== fuzz_heic_gpu 32"; timeout 900 python3 tools/fuzz_heic_gpu.py 32 2>&1 | grep -v amdgpu.ids | tail -2
} > gpurun_out/r03_fuzz2.log 2>&1
