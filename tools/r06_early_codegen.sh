#!/bin/bash
# r06 A/B on one box: chain.hip of the commit before the early CTU start against the current one, ring classes at full load
cd heif-decoder-lib_amd/csrc
cp chain.hip /tmp/chain_now.hip
for v in now before now before; do
  if [ $v = before ]; then cp ../../scratch/chain_before_early.hip chain.hip; else cp /tmp/chain_now.hip chain.hip; fi
  rm -f build/hip_chain.o; make >/dev/null 2>&1 || echo BUILD FAILED
  for c in 10bit_420_ctb32 8bit_420_ctb64 8bit_420_ctb32; do
    echo -n "$v $c 18432 tiles: "; (cd ../.. && HM_CLASS_TILES=18432 HM_CLASS_ONLY=$c python3 tools/bench_classes.py 2>/dev/null | tr -d '\n' | sed -E 's/.*k_recon_ms": ([0-9.]+).*/\1 ms/'); echo
  done
done
cp /tmp/chain_now.hip chain.hip; rm -f build/hip_chain.o; make >/dev/null 2>&1
