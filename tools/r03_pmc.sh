#!/bin/bash
# SQ counters of the bench kernels at 48 images (one launch = 2304 tiles); summary -> gpurun_out/r03_pmc_sq_$1.json
tag=${1:-b}
tools/pmc_sq.sh r03$tag --steps 2 --warmup 1 --images 48 > gpurun_out/r03_pmc_sq_$tag.log 2>&1
cp gpurun_out/pmc_r03$tag/summary.json gpurun_out/r03_pmc_sq_$tag.json
rm -rf gpurun_out/pmc_r03$tag
