#!/bin/bash
# r05 probe 3: where k_residual's vector instructions go (HM_R_SKIP: parts compiled out), counters per tile and kernel times
# (64 - no micro-ops - leaves k_chain without its control: counters only, never the bench)
export OBJ=residual KERNEL=k_residual
V=${V:-"-DHM_NONE|-DHM_R_SKIP=1|-DHM_R_SKIP=2|-DHM_R_SKIP=4|-DHM_R_SKIP=8|-DHM_R_SKIP=16|-DHM_R_SKIP=32"}
[ -n "$COUNTERS" ] && VARIANTS="$COUNTERS" MODE=counters PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" timeout 900 tools/probe_chain.sh
VARIANTS="$V" MODE=bench timeout 1200 tools/probe_chain.sh
