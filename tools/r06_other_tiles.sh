#!/bin/bash
# r06: the launcher's choice against the forced cuts for tiles of 256 x 256 and 1024 x 1024 (its thresholds were read from 512 x 512 tiles)
for t in 256 1024; do
  if [ $t = 256 ]; then counts="96 384 768 1536 3072 6144 24576"; else counts="6 24 96 384 768 1536"; fi
  echo "== tiles of $t x $t, 8-bit 4:2:0 CTB 32"; HM_CHECK_TILE=$t timeout 1500 python3 tools/check_launcher.py $counts 2>/dev/null | grep "tiles:"
done
for c in 8bit_420_ctb32 8bit_420_ctb16 10bit_420_ctb32; do echo "== 512 x 512, $c"; HM_CLASS_ONLY=$c timeout 900 python3 tools/check_launcher.py 192 384 512 768 1024 1280 1536 2048 2560 3072 2>/dev/null | grep tiles:; done
