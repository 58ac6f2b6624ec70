#!/bin/bash
python3 tools/tailf_probe.py 2>&1 | tail -4
for s in 1 2; do timeout 600 python3 tools/fuzz_tail.py $s 2>&1 | tail -2; done
