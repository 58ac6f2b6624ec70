#!/bin/bash
# r06: k_chain with two samples per lane for the angular 16x16 / 32x32 blocks of 8-bit pictures: kernel times and vector instructions per tile
VARIANTS="-DHM_NONE" OBJ=chain MODE=bench tools/probe_chain.sh
VARIANTS="-DHM_NONE" OBJ=chain MODE=counters KERNEL=k_chain PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" tools/probe_chain.sh
