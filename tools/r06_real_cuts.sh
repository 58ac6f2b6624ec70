#!/bin/bash
# r06: real 1080p intra pictures (32 per batch) and config 4 under forced cuts of k_chain against the launcher's own
run() { echo -n "$1: "; shift; env "$@" timeout 600 python3 tools/few_pictures_probe.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(v) for k, v in d['real_content_MP_per_s'].items()}, 'config4', round(d['config4_MP_per_s']), d['config4_kernels'].get('k_chain'))"; }
run "launcher" HM_NONE=1
run "a wave per chain" HM_CHAIN_PAIRS=3 HM_CHAIN_RING=0
run "a wave per chain, 16 per workgroup" HM_CHAIN_PAIRS=3 HM_CHAIN_RING=0 HM_CHAIN_NP=16
run "ring of 16 bands" HM_CHAIN_PAIRS=3 HM_CHAIN_RING=16
run "ring of 8 bands" HM_CHAIN_PAIRS=3 HM_CHAIN_RING=8
run "ring of 4 bands" HM_CHAIN_PAIRS=3 HM_CHAIN_RING=4
run "a wave per row" HM_CHAIN_PAIRS=2 HM_CHAIN_RING=0
