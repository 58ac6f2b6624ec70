#!/bin/bash
# r05: the 500-case parameter sweep (three stages against the oracle) with the launcher's own cuts and in forced ring / pair cuts,
# the fused tails' differential fuzz, and corrupted streams in two ring shapes
echo "== big_sweep, launcher's choice"; timeout 900 python3 tools/big_sweep.py 2>&1 | tail -2
echo "== big_sweep, ring of 3 row-pair waves"; HM_CHAIN_RING=3 HM_CHAIN_PAIRS=1 HM_QUAD_CLASS=1 timeout 900 python3 tools/big_sweep.py 2>&1 | tail -1
echo "== big_sweep, ring of 8 one-chain waves"; HM_CHAIN_RING=8 HM_CHAIN_PAIRS=3 HM_QUAD_CLASS=1 timeout 900 python3 tools/big_sweep.py 2>&1 | tail -1
echo "== fuzz_gpu, share 3"; HM_CHAIN_SHARE=3 timeout 900 python3 tools/fuzz_gpu.py 11 2>&1 | tail -1
echo "== fuzz_gpu, ring 5 rows"; HM_CHAIN_RING=5 HM_CHAIN_PAIRS=2 timeout 900 python3 tools/fuzz_gpu.py 12 2>&1 | tail -1
