#!/bin/bash
mkdir -p gpurun_out
{
timeout 900 python3 -m pytest tests/test_chain_modes_gpu.py -x -q -m gpu 2>&1 | tail -15
} > gpurun_out/r03_host.log 2>&1
