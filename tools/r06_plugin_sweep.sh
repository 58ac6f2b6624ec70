#!/bin/bash
# r06: the plugin path (48 tiles of one 12 MP grid through the decoder-plugin ABI, window of 8) by executors and linger of the shared device worker
for w in 2 4 6 8; do for l in 0 30 100; do
  echo -n "workers $w linger $l us: "; HM_PLUGIN_WORKERS=$w HM_PLUGIN_LINGER_US=$l timeout 300 python3 tools/plugin_probe.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_12MP_grid'], {k: v['ms_per_12MP_grid'] for k, v in d['wider_windows'].items()})"
done; done
echo -n "default: "; timeout 300 python3 tools/plugin_probe.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_12MP_grid'], {k: v['ms_per_12MP_grid'] for k, v in d['wider_windows'].items()})"
