#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python3 -m pytest tests/test_chain_modes_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python3 tools/few_pictures_probe.py 2>&1 | tail -1 | cut -c1-400
timeout 300 python3 tools/plugin_probe.py 2>&1 | tail -1 | cut -c1-160
timeout 600 python3 bench.py --mode grid --steps 10 --warmup 2 2>&1 | tail -1 | cut -c1-120
} > gpurun_out/r03_light.log 2>&1
