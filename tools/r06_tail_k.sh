#!/bin/bash
# r06: k_tail420 with K consecutive tiles per workgroup (the next tile's window loads in flight with this tile's pixel stores),
# tile heights, and the skeleton probes (HM_T_PROBE 7: no arithmetic; 39: + no stores; 71: + no loads) of the chosen variant.
VARIANTS="${VARIANTS:--DHM_TAIL_K=1|-DHM_TAIL_K=2|-DHM_TAIL_K=4|-DHM_TAIL_K=8|-DHM_TAIL_TH=96 -DHM_TAIL_MINW=2|-DHM_TAIL_TH=96 -DHM_TAIL_MINW=2 -DHM_TAIL_K=2|-DHM_TAIL_K=1 -DHM_T_PROBE=7|-DHM_TAIL_K=4 -DHM_T_PROBE=7|-DHM_TAIL_K=4 -DHM_T_PROBE=39|-DHM_TAIL_K=4 -DHM_T_PROBE=71}" OBJ=filters MODE=bench tools/probe_chain.sh
