#!/bin/bash
mkdir -p gpurun_out
{
echo "== chain modes (+ residual segments)"; timeout 1500 python3 -m pytest tests/test_chain_modes_gpu.py -x -q -m gpu 2>&1 | tail -3
echo "== decode / configs, default and forced segments"; timeout 900 python3 -m pytest tests/test_decode_gpu.py tests/test_configs_gpu.py tests/test_golden_heic.py tests/test_facade_gpu.py -x -q -m gpu 2>&1 | tail -2
HM_RESID_SEGS=3 timeout 900 python3 -m pytest tests/test_decode_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -2
echo "== few pictures"; python3 tools/few_pictures_probe.py 2>/dev/null | tail -1
echo "== plugin"; timeout 600 python3 tools/plugin_probe.py 2>&1 | tail -1 | cut -c1-420
echo "== bench quick"; timeout 600 python3 bench.py --quick --steps 5 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})"
} > gpurun_out/r03_segs.log 2>&1
