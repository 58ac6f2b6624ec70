#!/bin/bash
mkdir -p gpurun_out
{
for i in 1 2; do timeout 300 python3 tools/plugin_probe.py 2>&1 | tail -1 | cut -c1-160; done
echo "== plugin tests"; timeout 900 python3 -m pytest tests/test_facade_gpu.py tests/test_robustness.py -x -q -m gpu 2>&1 | tail -2
} > gpurun_out/r03_plugin.log 2>&1
