#!/bin/bash
q() { python3 bench.py --quick --no-parity --steps 10 --images $1 2>/tmp/err.txt | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('images', $1, {k.split('(')[0]: round(v['ms_per_step'],4) for k, v in d['kernels'].items()})"; grep "k_chain\]" /tmp/err.txt | sort | uniq -c | sort -rn | head -3; }
for e in 1 0; do
  for n in 1 2 3 4; do echo "== EARLY=$e images $n"; HM_CHAIN_DEBUG=1 HM_CHAIN_EARLY=$e q $n; done
done
for np in 4 8 16; do for e in 1 0; do echo "== EARLY=$e NP=$np images 2"; HM_CHAIN_NP=$np HM_CHAIN_EARLY=$e q 2; done; done
