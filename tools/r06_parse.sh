#!/bin/bash
# r06: host entropy decode per core on the GPU box's CPU (tools/parse_bench.py: hm_hevc_parse on the bench's tiles, one thread, best of 9 passes), five runs
grep -m1 "model name" /proc/cpuinfo
for i in 1 2 3 4 5; do python3 tools/parse_bench.py 48 9 2>&1 | tail -1; done
