#!/bin/bash
# A/B builds of chain.hip (OBJ=filters / residual: of that file) on the GPU box.  Every entry of VARIANTS is a set of extra compiler flags ("-DHM_Q_PROBE=n"
# compiles parts of the loop out - pictures wrong, parity gate off -: 1 no 4x4 path, 2 no wave-wide path, 4 wave-wide path
# without prediction; "-DHM_WPE=n" sets the waves per SIMD the register allocation aims for); the kernels are timed with
# bench.py.  usage (repo root): VARIANTS="-DHM_Q_PROBE=1|-DHM_Q_PROBE=2" tools/probe_chain.sh [bench args]
OBJ=${OBJ:-chain}
cd heif-decoder-lib_amd/csrc
IFS='|' read -ra VS <<< "${VARIANTS:--DHM_Q_PROBE=1|-DHM_Q_PROBE=2|-DHM_Q_PROBE=3}"
for v in "${VS[@]}"; do
  rm -f build/hip_$OBJ.o
  make HIPFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I. -I/opt/rocm/include $v" >/dev/null 2>&1
  echo -n "variant [$v]: "
  (cd ../.. && HM_CHAIN_DEBUG=1 python3 bench.py --quick --no-parity --steps 5 "$@" 2>/tmp/probe_err.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: v['ms_per_step'] for k, v in d['kernels'].items()})"; grep -m1 "k_chain" /tmp/probe_err.log || true)
done
rm -f build/hip_$OBJ.o; make >/dev/null 2>&1
