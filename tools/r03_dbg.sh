#!/bin/bash
# GPU-box run: first light of residual.hip + chain.hip
mkdir -p gpurun_out
{
echo "== debug tile512_a"; timeout 300 python3 tools/debug_recon.py tile512_a
echo "== debug ctb64"; timeout 300 python3 tools/debug_recon.py ctb64
echo "== debug mono8"; timeout 300 python3 tools/debug_recon.py mono8
echo "== debug hi422_10 (forced split)"; HM_QUAD_CLASS=1 timeout 300 python3 tools/debug_recon.py hi422_10
echo "== pytest decode (default classes)"; timeout 1200 python3 -m pytest tests/test_decode_gpu.py -x -q -m gpu 2>&1 | tail -15
echo "== pytest decode (all split)"; HM_QUAD_CLASS=1 timeout 1200 python3 -m pytest tests/test_decode_gpu.py -x -q -m gpu 2>&1 | tail -15
echo "== bench quick"; timeout 600 python3 bench.py --quick --steps 5 2>&1 | tail -3
echo "== bench quick old kernel"; HM_CHAIN=0 timeout 600 python3 bench.py --quick --steps 5 2>&1 | tail -3
} > gpurun_out/r03_dbg.log 2>&1
