#!/bin/bash
# Phase attribution of k_recon_quad (GPU box): rebuild recon_quad.hip with parts of the loop compiled out and time the
# kernel (the pictures are wrong, parity gate off).  usage (repo root): tools/probe_quad.sh
cd heif-decoder-lib_amd/csrc
for p in ${PROBES:-0 1 2 3}; do
  rm -f build/hip_recon_quad.o
  make HIPFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I. -I/opt/rocm/include -DHM_Q_PROBE=$p" >/dev/null 2>&1
  echo -n "probe $p (1: no 4x4 path, 2: no wave-wide path, 4: wave-wide without prediction, 8: without residual, 16: set-up only): "
  (cd ../.. && python3 bench.py --quick --no-parity --steps 5 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels']['k_recon']['ms_per_step'])")
done
rm -f build/hip_recon_quad.o; make >/dev/null 2>&1
