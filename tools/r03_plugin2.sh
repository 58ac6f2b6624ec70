#!/bin/bash
mkdir -p gpurun_out
HM_PLUGIN_DEBUG=1 HM_PLUGIN_LINGER_US=30 timeout 300 python3 tools/plugin_probe.py 2>&1 | tail -150 > gpurun_out/r03_plugin2.log
