#!/bin/bash
# whole GPU suite + default bench
mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests -x -q -m gpu > gpurun_out/r03_gputests.log 2>&1
tail -5 gpurun_out/r03_gputests.log
timeout 900 python3 bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
tail -c 3000 gpurun_out/r03_bench.json
