#!/bin/bash
# r06 A/B: LDS tile pitches of k_tail420 (8-bit headline and the 16-bit HDR class)
cd heif-decoder-lib_amd/csrc
for v in "-DHM_NONE=1" "-DHM_TAIL_LP=148 -DHM_TAIL_CP=84" "-DHM_TAIL_LP=152 -DHM_TAIL_CP=88" "-DHM_TAIL_LP=148" "-DHM_TAIL_LP=160 -DHM_TAIL_CP=96"; do
  rm -f build/hip_filters.o
  make HIPFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I. -I/opt/rocm/include $v" >/dev/null 2>&1 || echo "BUILD FAILED [$v]"
  echo "variant [$v]"
  (cd ../.. && python3 bench.py --quick --no-parity --steps 10 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  8-bit', {k.split('(')[0]: round(v['ms_per_step'],3) for k, v in d['kernels'].items()})"; python3 tools/tailf_probe.py 3 2>/dev/null | tail -1)
done
rm -f build/hip_filters.o; make >/dev/null 2>&1
