#!/bin/bash
for c in 8bit_420_ctb32 8bit_420_ctb16 10bit_420_ctb32; do echo "== $c"; HM_CLASS_ONLY=$c timeout 900 python3 tools/check_launcher.py 24 48 96 192 384 768 1536 2048 2>/dev/null | grep tiles; done
