#!/bin/bash
mkdir -p gpurun_out
{
OBJ=filters VARIANTS="-DHM_T_PROBE=0|-DHM_T_PROBE=1|-DHM_T_PROBE=2|-DHM_T_PROBE=4|-DHM_T_PROBE=7" tools/probe_chain.sh
OBJ=filters KERNEL=k_tail420 VARIANTS="-DHM_T_PROBE=0|-DHM_T_PROBE=1|-DHM_T_PROBE=2|-DHM_T_PROBE=4|-DHM_T_PROBE=7" tools/pmc_variants.sh
} > gpurun_out/r03_tailprobe.log 2>&1
