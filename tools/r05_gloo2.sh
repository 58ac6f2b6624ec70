#!/bin/bash
# the N > 1 host paths of bench.py on ONE GPU: two gloo ranks sharing it (functional run: barriers, max over ranks, per-rank legs, the
# config-3 share leg on every rank; grid mode with the padded gather) - what the driver's 8-GPU run exercises with nccl
export MASTER_ADDR=127.0.0.1
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --dist-backend gloo --allow-shared-gpu --images 24 --steps 3 --warmup 1 2>gpurun_out/gloo2_batch.err | tail -1 > gpurun_out/r05_bench_gloo2.json
python3 -c "
import json; d=json.loads(open('gpurun_out/r05_bench_gloo2.json').read()); print('batch mode:', d.get('value'), d.get('n_gpus'), d.get('config',{}).get('parity'), {k: (v if not isinstance(v, dict) else {q: v[q] for q in list(v)[:3]}) for k, v in d.items() if 'all_ranks' in k or k == 'error'})"
tail -3 gpurun_out/gloo2_batch.err
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --mode grid --dist-backend gloo --allow-shared-gpu --steps 2 --warmup 1 2>gpurun_out/gloo2_grid.err | tail -1 > gpurun_out/r05_bench_gloo2_grid.json
python3 -c "
import json; d=json.loads(open('gpurun_out/r05_bench_gloo2_grid.json').read()); print('grid mode:', d.get('value'), d.get('n_gpus'), d.get('config',{}).get('self_check'), d.get('gather'), d.get('error'))"
tail -3 gpurun_out/gloo2_grid.err
