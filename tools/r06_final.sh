#!/bin/bash
# r06: the round's committed measurements of the final code in one call (the caller copies gpurun_out/r06f_* into profiles/)
PMC_IMAGES=384 tools/gpu_round.sh r06f traffic
cp gpurun_out/pmc_traffic_r06f.json profiles/r06_pmc_traffic.json 2>/dev/null   # (bench.py's roofline.traffic looks it up: taken at the bench's own load)
PMC_IMAGES=384 tools/gpu_round.sh r06f sq stats tests smoke bench
bash tools/r06_tailf_counters.sh > gpurun_out/r06f_tailf_counters.txt 2>&1
{ echo "== tools/tailf_probe.py, the HDR class on k_tail420<16-bit> (stages 3 2 1 0 = deblocking + SAO / SAO / deblocking / neither)"; python3 tools/tailf_probe.py 2>/dev/null | tail -4
  echo "== the same on k_tailf (HM_TAIL_HDR16=0)"; HM_TAIL_HDR16=0 python3 tools/tailf_probe.py 2>/dev/null | tail -4; } >> gpurun_out/r06f_tailf_counters.txt
tools/gpu_round.sh r06f classes
python3 tools/check_launcher.py > gpurun_out/r06f_launcher_check.txt 2>&1
tail -5 gpurun_out/r06f_launcher_check.txt
# the few-pictures regime: the early CTU start A/B (knob for the cuts with a wave per chain / row, compile flag for the alternating ring), slabs of hm_decode_item
{ echo "== k_chain, early CTU start (tools/r06_early.sh)"; bash tools/r06_early.sh 2>&1 | grep -v "^== tests" | head -16
  echo "== hm_decode_item, one 12 MP grid, by tile rows per slab (0 = one batch behind the entropy decode)"; bash tools/r06_slabs.sh 2>&1; } > gpurun_out/r06f_few_pictures.txt 2>&1
TIMINGS="1 2 4" IMAGES=1 bash tools/r06_lat.sh > gpurun_out/r06f_chain_latency.txt 2>&1
