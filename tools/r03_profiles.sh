#!/bin/bash
# the round's profile set: rocprofv3 kernel stats of the default bench command, TCC traffic, SQ counters
export TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/r03_prof; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/$out --output-format csv -- python3 $OLDPWD/bench.py --quick > $OLDPWD/gpurun_out/r03_bench_under_rocprof.json 2> $OLDPWD/gpurun_out/r03_bench_under_rocprof.err)
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03_bench_kernel_stats.csv
tools/pmc_traffic.sh r03 48 > gpurun_out/r03_pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic_r03.json gpurun_out/r03_pmc_traffic.json
tools/r03_pmc.sh final
rm -rf gpurun_out/pmc_traffic_r03 $out
