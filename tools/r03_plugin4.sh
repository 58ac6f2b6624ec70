#!/bin/bash
mkdir -p gpurun_out
{ for i in 1 2; do timeout 600 python3 tools/plugin_probe.py 2>&1 | tail -1 | cut -c1-700; done; } > gpurun_out/r03_plugin4.log 2>&1
