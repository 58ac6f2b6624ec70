#!/bin/bash
# r06: where the early CTU start stops paying - the classes whose waves hand rows over at every picture count (rings), reconstruction ms by tile count, knob chain_early 1 / 0
for c in 10bit_420_ctb32 8bit_420_ctb64 10bit_422_ctb32 8bit_420_ctb32; do
  for n in 1536 3072 4608 9216 18432; do
    for e in 1 0; do
      echo -n "$c $n tiles early=$e: "; HM_CHAIN_EARLY=$e HM_CLASS_TILES=$n HM_CLASS_ONLY=$c python3 tools/bench_classes.py 2>/dev/null | tr -d '\n' | sed -E 's/.*k_recon_ms": ([0-9.]+).*/\1 ms/'; echo
    done
  done
done
