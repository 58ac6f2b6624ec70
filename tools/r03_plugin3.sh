#!/bin/bash
# the plugin path with the phases of every batch of the device worker, and the kernels of those small batches
export TMPDIR=/tmp
mkdir -p gpurun_out
{
HM_PLUGIN_DEBUG=1 HM_CHAIN_DEBUG=1 timeout 300 python3 tools/plugin_probe.py 2>&1 | grep -v "amdgpu.ids" | tail -70 | cut -c1-200
echo "== kernels of the plugin path"
rm -rf /tmp/pl_prof; rocprofv3 --kernel-trace --stats -d /tmp/pl_prof --output-format csv -- python3 tools/plugin_probe.py > /tmp/pl_prof.log 2>&1
f=$(find /tmp/pl_prof -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200
} > gpurun_out/r03_plugin3.log 2>&1
