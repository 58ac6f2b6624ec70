#!/bin/bash
# r06: k_tail420 with a lane per half unit in the filter passes: time A/B and SQ counters per tile
VARIANTS="-DHM_TAIL_HALF_UNITS=0|-DHM_TAIL_HALF_UNITS=1" OBJ=filters MODE=bench tools/probe_chain.sh
VARIANTS="-DHM_TAIL_HALF_UNITS=0|-DHM_TAIL_HALF_UNITS=1|-DHM_TAIL_HALF_UNITS=1 -DHM_T_PROBE=1" OBJ=filters MODE=counters KERNEL=k_tail420 PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" tools/probe_chain.sh
