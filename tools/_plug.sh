mkdir -p gpurun_out
{
for w in 3 4 6 8; do for l in 0 30 100; do
echo -n "workers $w linger $l: "; HM_PLUGIN_WORKERS=$w HM_PLUGIN_LINGER_US=$l timeout 300 python3 tools/plugin_probe.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_12MP_grid'], d['hm_decode_item_ms'], d['ratio_to_hm_decode_item'], {k:(v['ms_per_12MP_grid'],v['ratio_to_hm_decode_item']) for k,v in d['wider_windows'].items()})"
done; done
echo "== quick (tail UNI)"; timeout 600 python3 bench.py --quick --steps 10 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})"
echo "== tail tests"; timeout 900 python3 -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "fused or config2 or grouped" 2>&1 | tail -2
} > gpurun_out/r04i_plug.log 2>&1
cat gpurun_out/r04i_plug.log
