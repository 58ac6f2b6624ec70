# round-2 evidence run (GPU box, repo root): full GPU test suite, default bench, rocprofv3 kernel stats, TCC traffic, SQ counters
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -n 4 2>&1 | tail -4 > gpurun_out/r02_gputest_full.log
timeout 1200 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
tail -c 400 gpurun_out/r02_bench_default.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r02_prof_stats --output-format csv -- python3 bench.py --no-parity --quick > gpurun_out/r02_bench_under_rocprof.json 2> gpurun_out/r02_bench_under_rocprof.err
HM_TAIL_FUSED=0 rocprofv3 --kernel-trace --stats -d gpurun_out/r02_prof_stats_unfused --output-format csv -- python3 bench.py --no-parity --quick > gpurun_out/r02_bench_under_rocprof_unfused.json 2> gpurun_out/r02_bench_under_rocprof_unfused.err
find gpurun_out/r02_prof_stats gpurun_out/r02_prof_stats_unfused -name "*kernel_stats.csv" | head -4
bash tools/pmc_traffic.sh r02 384 2>&1 | tail -40
bash tools/pmc_sq.sh r02 --steps 2 --warmup 1 --images 384 2>&1 | tail -60
cat gpurun_out/r02_gputest_full.log
