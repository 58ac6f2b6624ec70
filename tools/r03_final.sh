#!/bin/bash
# end of round: whole GPU suite, default bench, 2-rank bench on one GPU (gloo, functional test of the N > 1 path)
mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests -x -q -m gpu > gpurun_out/r03_gputests.log 2>&1
tail -3 gpurun_out/r03_gputests.log
timeout 900 python3 bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --dist-backend gloo --allow-shared-gpu --steps 5 --warmup 1 --images 96 > gpurun_out/r03_bench_gloo2.json 2> gpurun_out/r03_bench_gloo2.err
tail -c 400 gpurun_out/r03_bench_gloo2.json
