#!/bin/bash
mkdir -p gpurun_out
{
echo "== pytest decode (all split)"; HM_QUAD_CLASS=1 timeout 900 python3 -m pytest tests/test_decode_gpu.py -x -q -m gpu 2>&1 | tail -2
echo "== pytest decode + configs + chain modes (default)"; timeout 1200 python3 -m pytest tests/test_decode_gpu.py tests/test_configs_gpu.py tests/test_chain_modes_gpu.py -x -q -m gpu 2>&1 | tail -2
echo "== bench quick"; HM_CHAIN_DEBUG=1 timeout 600 python3 bench.py --quick --steps 5 2>/tmp/e.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})"; grep -m1 "k_chain" /tmp/e.log
timeout 600 python3 tools/few_pictures_probe.py 2>&1 | tail -1 | cut -c1-300
} > gpurun_out/r03_q.log 2>&1
