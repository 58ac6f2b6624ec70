#!/bin/bash
# r05 validation of the round's kernels: differential fuzz of the fused tails (3 seeds) and of the reconstruction kernels, then the whole GPU suite
for s in 1 2 3; do echo "== fuzz_tail seed $s"; timeout 600 python3 tools/fuzz_tail.py $s 2>&1 | tail -3; done
echo "== fuzz_gpu"; timeout 900 python3 tools/fuzz_gpu.py 7 2>&1 | tail -3
