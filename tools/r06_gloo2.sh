#!/bin/bash
# the N > 1 host paths of bench.py on ONE GPU: two gloo ranks sharing it (functional run: barriers, max over ranks, per-rank legs, the
# config-3 share leg on every rank; grid mode with the pipelined point-to-point gather into the root's rows - whole slabs and chunks of
# one tile row - and the shared host image) - what the driver's 8-GPU run exercises with nccl
export MASTER_ADDR=127.0.0.1
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --dist-backend gloo --allow-shared-gpu --images 24 --steps 3 --warmup 1 2>gpurun_out/gloo2_batch.err | tail -1 > gpurun_out/r06_bench_gloo2.json
python3 -c "
import json; d=json.loads(open('gpurun_out/r06_bench_gloo2.json').read()); print('batch mode:', d.get('value'), d.get('n_gpus'), d.get('config',{}).get('parity'), {k: (v if not isinstance(v, dict) else {q: v[q] for q in list(v)[:3]}) for k, v in d.items() if 'all_ranks' in k or k == 'error'})"
tail -3 gpurun_out/gloo2_batch.err
for chunk in 0 4; do
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2961$((2 + chunk)) bench.py --gpus 2 --mode grid --dist-backend gloo --allow-shared-gpu --steps 4 --warmup 1 --grid-chunk-rows $chunk 2>gpurun_out/gloo2_grid$chunk.err | tail -1 > gpurun_out/r06_bench_gloo2_grid_chunk$chunk.json
python3 -c "
import json; d=json.loads(open('gpurun_out/r06_bench_gloo2_grid_chunk$chunk.json').read()); print('grid mode, chunk $chunk:', d.get('value'), d.get('n_gpus'), d.get('config',{}).get('self_check'), d.get('config',{}).get('chunks_per_rank'), d.get('k_only'), d.get('single_grid_latency'), d.get('gather'), d.get('host_gather'), d.get('error'))"
tail -3 gpurun_out/gloo2_grid$chunk.err
done
echo "== one rank, grid mode"
python3 bench.py --mode grid --steps 5 --warmup 1 2>gpurun_out/grid1.err | tail -1 > gpurun_out/r06_bench_grid1.json
python3 -c "
import json; d=json.loads(open('gpurun_out/r06_bench_grid1.json').read()); print(d.get('value'), d.get('config',{}).get('self_check'), d.get('k_only'), d.get('host_gather'), d.get('error'))"
tail -3 gpurun_out/grid1.err
