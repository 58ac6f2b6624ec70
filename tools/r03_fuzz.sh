#!/bin/bash
# end-of-round differential fuzzing of the r03 kernels: corrupted-but-parsable streams (GPU vs oracle), in the default cut and
# with the split-chain path forced for every class and every cut of a picture; mutated HEIC files through the image path
mkdir -p gpurun_out
{
for seed in 301 302; do echo "== fuzz_gpu seed $seed"; timeout 900 python3 tools/fuzz_gpu.py $seed 2>&1 | tail -2; done
for e in "HM_CHAIN_PAIRS=0" "HM_CHAIN_PAIRS=1" "HM_CHAIN_PAIRS=3" "HM_CHAIN_SHARE=3"; do
  echo "== fuzz_gpu seed 303, all classes on the split-chain path, $e"; env HM_QUAD_CLASS=1 $e timeout 900 python3 tools/fuzz_gpu.py 303 2>&1 | tail -2
done
echo "== fuzz_heic_gpu"; timeout 900 python3 tools/fuzz_heic_gpu.py 31 2>&1 | tail -3
} > gpurun_out/r03_fuzz.log 2>&1
