#!/bin/bash
# r06 validation of the round's kernels (two samples per lane in k_chain's 16x16 / 32x32 angular blocks, packed 16-bit deblocking,
# k_tailf with records / block map through LDS, 64-row tiles and the packed integer matrix): the 500-case parameter sweep (three stages
# against the oracle) with the launcher's own cuts and in two forced ring cuts, corrupted streams in two cuts, the fused tails' fuzz
echo "== big_sweep, launcher's choice"; timeout 900 python3 tools/big_sweep.py 2>&1 | tail -2
echo "== big_sweep, ring of 3 row-pair waves"; HM_CHAIN_RING=3 HM_CHAIN_PAIRS=1 HM_QUAD_CLASS=1 timeout 900 python3 tools/big_sweep.py 2>&1 | tail -1
echo "== big_sweep, ring of 8 one-chain waves"; HM_CHAIN_RING=8 HM_CHAIN_PAIRS=3 HM_QUAD_CLASS=1 timeout 900 python3 tools/big_sweep.py 2>&1 | tail -1
echo "== fuzz_gpu"; timeout 900 python3 tools/fuzz_gpu.py 7 2>&1 | tail -2
echo "== fuzz_gpu, share 3"; HM_CHAIN_SHARE=3 timeout 900 python3 tools/fuzz_gpu.py 11 2>&1 | tail -1
echo "== fuzz_gpu, ring 5 rows"; HM_CHAIN_RING=5 HM_CHAIN_PAIRS=2 timeout 900 python3 tools/fuzz_gpu.py 12 2>&1 | tail -1
for s in 1 2 3; do echo "== fuzz_tail seed $s"; timeout 600 python3 tools/fuzz_tail.py $s 2>&1 | tail -2; done
