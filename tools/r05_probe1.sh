#!/bin/bash
# r05 probe 1: the baseline of the round on this box + VALU instruction counts of k_chain per tile with parts compiled out
# (HM_Q_PROBE bits: 1 no 4x4 pass, 2 no wave-wide path; HM_D_SKIP: mask of wave-wide block classes that are skipped -
#  1 F8A, 2 F8O, 3 S8, 4 I16, 5 B4, 7 / 8 / 9 the general path for 8x8 / 16x16 / 32x32)
python3 bench.py --quick --steps 10 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('baseline', d['value'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})"
VARIANTS="-DHM_NONE|-DHM_Q_PROBE=1|-DHM_Q_PROBE=2|-DHM_D_SKIP=0x2|-DHM_D_SKIP=0x4|-DHM_D_SKIP=0x8|-DHM_D_SKIP=0x10|-DHM_D_SKIP=0x20|-DHM_D_SKIP=0x180|-DHM_D_SKIP=0x200" MODE=counters PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" tools/probe_chain.sh
