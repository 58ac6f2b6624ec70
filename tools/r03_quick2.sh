#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python3 -m pytest tests/test_decode_gpu.py -x -q -m gpu -k "large_single or synth_corpus" 2>&1 | tail -3
timeout 600 python3 bench.py --quick --steps 5 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})"
} > gpurun_out/r03_quick2.log 2>&1
