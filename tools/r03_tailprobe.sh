#!/bin/bash
mkdir -p gpurun_out
V=${V:-"-DHM_T_PROBE=0|-DHM_T_PROBE=8|-DHM_T_PROBE=16|-DHM_T_PROBE=24"}
{
OBJ=filters KERNEL=k_tail420 VARIANTS="$V" tools/pmc_variants.sh
} > gpurun_out/r03_tailprobe.log 2>&1
