#!/bin/bash
# K clock with the grouped multi-stream execute: the fused tail of one group of images under the reconstruction of another
mkdir -p gpurun_out
run() { env "$@" python bench.py --quick --steps 6 --warmup 2 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d['config']['parity'][:9])"; }
{
run HM_K_GROUPS=0
run HM_K_GROUPS=2 HM_K_STREAMS=2
run HM_K_GROUPS=2 HM_K_STREAMS=2 HM_K_PRIO=1
run HM_K_GROUPS=2 HM_K_STREAMS=2 HM_K_SPLIT=60
run HM_K_GROUPS=2 HM_K_STREAMS=2 HM_K_SPLIT=70
run HM_K_GROUPS=2 HM_K_STREAMS=2 HM_K_SPLIT=70 HM_K_PRIO=1
run HM_K_GROUPS=2 HM_K_STREAMS=2 HM_K_SPLIT=40
run HM_K_GROUPS=3 HM_K_STREAMS=3
run HM_K_GROUPS=3 HM_K_STREAMS=3 HM_K_PRIO=1
run HM_K_GROUPS=4 HM_K_STREAMS=4
run HM_K_GROUPS=4 HM_K_STREAMS=4 HM_K_PRIO=1
run HM_K_GROUPS=8 HM_K_STREAMS=8
} | tee gpurun_out/k_groups.txt
