#!/bin/bash
q() { python3 bench.py --quick --no-parity --steps 10 --images $1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('images', $1, {k.split('(')[0]: round(v['ms_per_step'],4) for k, v in d['kernels'].items()})"; }
for n in 1 2; do
echo "== auto"; q $n
for np in 4 8 12 16; do echo "== chains NP=$np"; HM_CHAIN_RING=0 HM_CHAIN_PAIRS=3 HM_CHAIN_NP=$np q $n; done
echo "== ring8 chains"; HM_CHAIN_RING=8 HM_CHAIN_PAIRS=3 q $n
echo "== ring4 chains"; HM_CHAIN_RING=4 HM_CHAIN_PAIRS=3 q $n
echo "== rows NP=8"; HM_CHAIN_RING=0 HM_CHAIN_PAIRS=2 HM_CHAIN_NP=8 q $n
echo "== rows NP=16"; HM_CHAIN_RING=0 HM_CHAIN_PAIRS=2 HM_CHAIN_NP=16 q $n
done
