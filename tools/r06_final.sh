#!/bin/bash
# r06: the round's committed measurements of the final code in one call (the caller copies gpurun_out/r06f_* into profiles/)
PMC_IMAGES=384 tools/gpu_round.sh r06f traffic
cp gpurun_out/pmc_traffic_r06f.json profiles/r06_pmc_traffic.json 2>/dev/null   # (bench.py's roofline.traffic looks it up: taken at the bench's own load)
PMC_IMAGES=384 tools/gpu_round.sh r06f sq stats tests smoke bench
bash tools/r06_tailf_counters.sh > gpurun_out/r06f_tailf_counters.txt 2>&1
python3 tools/tailf_probe.py 2>/dev/null | tail -4 >> gpurun_out/r06f_tailf_counters.txt
tools/gpu_round.sh r06f classes
python3 tools/check_launcher.py > gpurun_out/r06f_launcher_check.txt 2>&1
tail -5 gpurun_out/r06f_launcher_check.txt
