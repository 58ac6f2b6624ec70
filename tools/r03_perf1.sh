#!/bin/bash
mkdir -p gpurun_out
{
echo "== bench quick"; timeout 600 python3 bench.py --quick --steps 5 2>&1 | tail -1
PROBES="1 2 3" tools/probe_chain.sh
} > gpurun_out/r03_perf1.log 2>&1
tools/pmc_sq.sh r03a --steps 2 --warmup 1 --images 48 > gpurun_out/r03_pmc_sq_a.log 2>&1
cp gpurun_out/pmc_r03a/summary.json gpurun_out/r03_pmc_sq_a.json
