#!/bin/bash
# same binary (chain.hip built for 5 waves per SIMD), different workgroup sizes = different waves per CU
make -C heif-decoder-lib_amd/csrc -B build/hip_chain.o HIPFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I. -I/opt/rocm/include -DHM_WPE=5" > /dev/null 2>&1
make -C heif-decoder-lib_amd/csrc > /dev/null 2>&1
for np in 4 10 9 1; do
  echo -n "np=$np: "; HM_CHAIN_NP=$np HM_CHAIN_DEBUG=1 python3 bench.py --quick --no-parity --steps 5 2>/tmp/e.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: v['ms_per_step'] for k, v in d['kernels'].items()})"; grep -m1 k_chain /tmp/e.log
done
