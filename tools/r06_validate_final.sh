#!/bin/bash
# r06: every validation sweep of the round once more on the final code
bash tools/r06_validate.sh 2>&1 | grep -v amdgpu.ids
bash tools/r06_early4.sh 2>&1 | grep -v amdgpu.ids
echo "== tools/stress_cuts.py, 100 repetitions"; timeout 1200 python3 tools/stress_cuts.py 100 2>&1 | tail -1
