for w in 2 4 8; do
  echo "== HM_QUAD_WAVES=$w"
  HM_QUAD_WAVES=$w timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | tail -3
done
echo "== auto"
timeout 900 python -m pytest tests/test_decode_gpu.py tests/test_configs_gpu.py tests/test_golden_heic.py -x -q 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/r02_bench_mw.json 2> gpurun_out/r02_bench_mw.err
tail -c 300 gpurun_out/r02_bench_mw.err
