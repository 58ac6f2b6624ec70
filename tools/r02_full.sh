export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r02_gputest_full.log
timeout 900 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
tail -c 600 gpurun_out/r02_bench_default.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r02_prof_stats --output-format csv -- python3 bench.py --no-parity --quick > gpurun_out/r02_bench_under_rocprof.json 2> gpurun_out/r02_bench_under_rocprof.err
find gpurun_out/r02_prof_stats -name "*kernel_stats.csv" | head -2
bash tools/pmc_traffic.sh r02 384 2>&1 | tail -30
cat gpurun_out/r02_gputest_full.log
