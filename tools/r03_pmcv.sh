#!/bin/bash
mkdir -p gpurun_out
VARIANTS="${V:--DHM_Q_PROBE=0|-DHM_Q_PROBE=1|-DHM_Q_PROBE=2|-DHM_Q_PROBE=3|-DHM_Q_PROBE=8|-DHM_Q_PROBE=128|-DHM_Q_PROBE=256}" tools/pmc_variants.sh > gpurun_out/r03_pmcv.log 2>&1
