#!/bin/bash
# r05: the staircase of the wave-per-picture cut (8-bit 4:2:0 CTB 32, ms of both reconstruction kernels by tile count) with the partial
# last round in one launch (HM_CHAIN_SPLIT=0) and as a launch of its own beside the full rounds (1: queued first, 2: queued second)
for split in 0 1 2; do
  echo "== HM_CHAIN_SPLIT=$split"
  for n in ${TILES:-4096 5120 5376 5632 6144 6400 10752 11264}; do
    echo -n "$n tiles: "; HM_CHAIN_SPLIT=$split HM_CLASS_TILES=$n HM_CLASS_ONLY=8bit_420_ctb32 python3 tools/bench_classes.py 2>/dev/null | tr -d '\n' | sed -E 's/.*k_recon_ms": ([0-9.]+).*/\1 ms/'; echo
  done
done
echo "== correctness: 5632 copies of a 512x512 tile against the oracle, the launcher's own choice"
HM_CHECK_COPIES=5632 HM_CHAIN_DEBUG=1 PYTHONPATH=.:tests timeout 900 python3 tests/chain_mode_check.py tile512_a 2>&1 | tail -6
