#!/bin/bash
bash tools/r06_early.sh 2>&1 | grep -v "^\.\.\." | head -16
for c in 10bit_420_ctb32 8bit_420_ctb64; do echo -n "$c 18432 tiles: "; HM_CLASS_TILES=18432 HM_CLASS_ONLY=$c python3 tools/bench_classes.py 2>/dev/null | tr -d '\n' | sed -E 's/.*k_recon_ms": ([0-9.]+).*/\1 ms/'; echo; done
python3 tools/tailf_probe.py 3 2>/dev/null | tail -1
timeout 600 python3 tools/stress_cuts.py 30 2>&1 | tail -1
bash tools/r06_early4.sh 2>&1 | grep -A1 "^=="
