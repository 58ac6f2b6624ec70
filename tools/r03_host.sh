#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python3 -m pytest tests/test_pipeline_gpu.py tests/test_transforms.py tests/test_facade_gpu.py tests/test_golden_heic.py -x -q -m gpu 2>&1 | tail -5
echo "== bench default"; timeout 900 python3 bench.py --steps 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; tail -c 400 gpurun_out/r03_bench.err
echo "== bench 2 ranks (gloo, shared GPU)"; timeout 900 python3 bench.py --gpus 2 --allow-shared-gpu --dist-backend gloo --images 48 --steps 3 > gpurun_out/r03_bench_gloo2.json 2> gpurun_out/r03_bench_gloo2.err; tail -c 600 gpurun_out/r03_bench_gloo2.err
} > gpurun_out/r03_host.log 2>&1
