#!/bin/bash
mkdir -p gpurun_out
{
echo "== pytest configs/transforms/facade/golden"; timeout 1500 python3 -m pytest tests/test_configs_gpu.py tests/test_transforms.py tests/test_facade_gpu.py tests/test_golden_heic.py tests/test_colour_gpu.py -x -q -m gpu 2>&1 | tail -5
echo "== bench quick (with parity)"; timeout 600 python3 bench.py --quick --steps 5 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})"
} > gpurun_out/r03_tail.log 2>&1
