#!/bin/bash
mkdir -p gpurun_out
OBJ=${OBJ:-chain} VARIANTS="$V" tools/probe_chain.sh 2>&1 | grep -v "^\[k_chain" > gpurun_out/r03_ab.log
