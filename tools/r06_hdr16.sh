#!/bin/bash
# r06: the HDR class (10-bit 4:2:0 -> RGB24 / RGBA32) on k_tail420's 16-bit instantiation against k_tailf (HM_TAIL_HDR16=0)
echo "== probe, k_tail420<uint16_t>"; python3 tools/tailf_probe.py 2>&1 | tail -4
echo "== probe, k_tailf"; HM_TAIL_HDR16=0 python3 tools/tailf_probe.py 2>&1 | tail -4
for s in 1 2 3; do echo "== fuzz_tail seed $s"; timeout 600 python3 tools/fuzz_tail.py $s 2>&1 | tail -6; done
echo "== tests"; timeout 1500 python3 -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "float_tail" 2>&1 | tail -5
