#!/bin/bash
for np in 0 8 6; do HM_CHAIN_NP=$np python3 tools/corun_probe.py 2>&1 | tail -3; done
