#!/bin/bash
for r in 0 -1 2 3; do echo "== HM_GRID_SLAB_ROWS=$r"; HM_GRID_SLAB_ROWS=$r python3 tools/decode_latency.py 16 8 2>/dev/null | tail -2; done
