#!/bin/bash
# host entropy decode after a parser change: the GPU suites that compare pictures of every class with the oracle, the
# parse rate on this box's CPU, the plugin path
mkdir -p gpurun_out
{
lscpu | grep -i "model name\|^CPU(s)\|MHz" | head -4
echo "== parse rate"; for i in 1 2 3; do python3 tools/parse_bench.py 48 7; done
echo "== pytest"; timeout 2400 python3 -m pytest tests/test_decode_gpu.py tests/test_configs_gpu.py tests/test_transforms.py tests/test_facade_gpu.py tests/test_golden_heic.py -x -q -m gpu 2>&1 | tail -3
echo "== plugin"; timeout 600 python3 tools/plugin_probe.py 2>&1 | tail -6
} > gpurun_out/r03_parse.log 2>&1
