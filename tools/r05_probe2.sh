#!/bin/bash
# r05 probe 2: SQ counters of k_tail420 per 512x512 tile (thousands), whole kernel and with parts compiled out (HM_T_PROBE 1: no deblocking, 2: no SAO, 4: no matrix)
VARIANTS="${VARIANTS:--DHM_NONE|-DHM_T_PROBE=1}" OBJ=filters KERNEL=k_tail420 MODE=counters PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" tools/probe_chain.sh
