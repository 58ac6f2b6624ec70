#!/bin/bash
# r06: the early CTU start in the cuts the launcher does not take by itself
echo "== big_sweep, ring of 4 one-chain waves that keep their kind"; HM_CHAIN_RING=4 HM_CHAIN_PAIRS=3 HM_CHAIN_ALT=0 HM_QUAD_CLASS=1 timeout 900 python3 tools/big_sweep.py 2>&1 | tail -1
echo "== big_sweep, ring of 2 one-chain waves that keep their kind (old rule)"; HM_CHAIN_RING=2 HM_CHAIN_PAIRS=3 HM_CHAIN_ALT=0 HM_QUAD_CLASS=1 timeout 900 python3 tools/big_sweep.py 2>&1 | tail -1
echo "== big_sweep, ring of 2 alternating one-chain waves"; HM_CHAIN_RING=2 HM_CHAIN_PAIRS=3 HM_QUAD_CLASS=1 timeout 900 python3 tools/big_sweep.py 2>&1 | tail -1
echo "== big_sweep, ring of 4 rows"; HM_CHAIN_RING=4 HM_CHAIN_PAIRS=2 HM_QUAD_CLASS=1 timeout 900 python3 tools/big_sweep.py 2>&1 | tail -1
echo "== big_sweep, a wave per chain, 2 waves per workgroup (hand-over through HBM every row)"; HM_CHAIN_RING=0 HM_CHAIN_PAIRS=3 HM_CHAIN_NP=2 HM_QUAD_CLASS=1 timeout 900 python3 tools/big_sweep.py 2>&1 | tail -1
echo "== big_sweep, a wave per row, 3 waves per workgroup"; HM_CHAIN_RING=0 HM_CHAIN_PAIRS=2 HM_CHAIN_NP=3 HM_QUAD_CLASS=1 timeout 900 python3 tools/big_sweep.py 2>&1 | tail -1
echo "== fuzz_gpu, a wave per chain, 2 waves per workgroup"; HM_CHAIN_RING=0 HM_CHAIN_PAIRS=3 HM_CHAIN_NP=2 timeout 900 python3 tools/fuzz_gpu.py 13 2>&1 | tail -1
echo "== fuzz_gpu, ring of 16 alternating"; HM_CHAIN_RING=16 HM_CHAIN_PAIRS=3 timeout 900 python3 tools/fuzz_gpu.py 14 2>&1 | tail -1
echo "== big_sweep, ring of 8 alternating one-chain waves WITHOUT the early start (the kernels of saturated launches)"; HM_CHAIN_EARLY=0 HM_CHAIN_RING=8 HM_CHAIN_PAIRS=3 HM_QUAD_CLASS=1 timeout 900 python3 tools/big_sweep.py 2>&1 | tail -1
echo "== big_sweep, a wave per chain WITHOUT the early start"; HM_CHAIN_EARLY=0 HM_CHAIN_RING=0 HM_CHAIN_PAIRS=3 HM_QUAD_CLASS=1 timeout 900 python3 tools/big_sweep.py 2>&1 | tail -1
