#!/bin/bash
mkdir -p gpurun_out
{
for m in 1 2 3; do
echo "== pytest decode (all split, HM_CHAIN_PAIRS=$m)"; HM_CHAIN_PAIRS=$m HM_QUAD_CLASS=1 timeout 900 python3 -m pytest tests/test_decode_gpu.py -x -q -m gpu 2>&1 | tail -3
done
echo "== pytest decode (default)"; timeout 900 python3 -m pytest tests/test_decode_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "== pytest decode (all split, no pairs)"; HM_CHAIN_PAIRS=0 HM_QUAD_CLASS=1 timeout 900 python3 -m pytest tests/test_decode_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "== pytest configs"; timeout 900 python3 -m pytest tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -3
} > gpurun_out/r03_pairs.log 2>&1
