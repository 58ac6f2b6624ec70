#!/bin/bash
# r06: the few-pictures cuts of k_chain with the early CTU start (a CTU starts when the CTU above is done; OP_FAR blocks wait for the one above-right)
# against the rule before (HM_CHAIN_EARLY=0): reconstruction ms for 1 / 2 / 4 / 8 / 32 images of 48 tiles, config 4, real content
q() { python3 bench.py --quick --no-parity --steps 10 --images $1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('images', $1, 'K MP/s', d['value'], {k.split('(')[0]: round(v['ms_per_step'],4) for k, v in d['kernels'].items()})"; }
for e in 1 0; do
  echo "== HM_CHAIN_EARLY=$e"
  for n in 1 2 4 8 32; do HM_CHAIN_EARLY=$e q $n; done
  HM_CHAIN_EARLY=$e timeout 600 python3 tools/few_pictures_probe.py 2>/dev/null | tail -1
done
echo "== tests"; timeout 1500 python3 -m pytest tests/test_chain_modes_gpu.py tests/test_decode_gpu.py -m gpu -x -q 2>&1 | tail -4
