#!/bin/bash
mkdir -p gpurun_out
{
echo "== debug tile512_a (pairs)"; timeout 120 python3 tools/debug_recon.py tile512_a 2>&1 | tail -8
echo "== debug ctb64_wpp (pairs)"; timeout 120 python3 tools/debug_recon.py ctb64_wpp 2>&1 | tail -8
echo "== debug mono8 (pairs)"; timeout 120 python3 tools/debug_recon.py mono8 2>&1 | tail -5
echo "== pytest decode (default)"; timeout 900 python3 -m pytest tests/test_decode_gpu.py -x -q -m gpu 2>&1 | tail -4
echo "== pytest decode (all split, pairs)"; HM_QUAD_CLASS=1 timeout 900 python3 -m pytest tests/test_decode_gpu.py -x -q -m gpu 2>&1 | tail -4
echo "== pytest decode (all split, no pairs)"; HM_CHAIN_PAIRS=0 HM_QUAD_CLASS=1 timeout 900 python3 -m pytest tests/test_decode_gpu.py -x -q -m gpu 2>&1 | tail -4
echo "== pytest configs"; timeout 900 python3 -m pytest tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -4
} > gpurun_out/r03_pairs.log 2>&1
