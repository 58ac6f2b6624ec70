#!/bin/bash
# Phase attribution of k_chain (GPU box): rebuild chain.hip with parts of the loop compiled out and time the kernels
# (the pictures are wrong, parity gate off).  usage (repo root): PROBES="0 1 2 3" tools/probe_chain.sh [bench args]
cd heif-decoder-lib_amd/csrc
for p in ${PROBES:-0 1 2 3}; do
  rm -f build/hip_chain.o
  make HIPFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I. -I/opt/rocm/include -DHM_Q_PROBE=$p" >/dev/null 2>&1
  echo -n "probe $p (1: no 4x4 path, 2: no wave-wide path, 4: wave-wide without prediction): "
  (cd ../.. && python3 bench.py --quick --no-parity --steps 5 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: v['ms_per_step'] for k, v in d['kernels'].items()})")
done
rm -f build/hip_chain.o; make >/dev/null 2>&1
