#!/bin/bash
# mid-size batches: a wave per picture against W waves per picture that take its row pairs in turn (kernel ms per step)
mkdir -p gpurun_out
{
for w in 2 4 7; do
echo "== correctness (HM_CHAIN_SHARE=$w, all split-chain classes)"; HM_CHAIN_SHARE=$w HM_QUAD_CLASS=1 timeout 900 python3 -m pytest tests/test_decode_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -2
done
for n in 6 8 12 22 32 43; do
  for e in "HM_X=1" "HM_CHAIN_PAIRS=0" "HM_CHAIN_PAIRS=1" "HM_CHAIN_SHARE=2" "HM_CHAIN_SHARE=4" "HM_CHAIN_SHARE=7"; do
    echo -n "images $n [$e]: "
    env $e HM_CHAIN_DEBUG=1 timeout 300 python3 bench.py --quick --no-parity --steps 5 --images $n 2>/tmp/e.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: v['ms_per_step'] for k, v in d['kernels'].items()})" 2>&1 | tail -1
    grep -m1 "k_chain" /tmp/e.log | cut -c1-110
  done
done
} > gpurun_out/r03_share.log 2>&1
