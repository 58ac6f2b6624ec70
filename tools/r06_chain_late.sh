#!/bin/bash
# r06, VERDICT r05 item 2a: k_chain with the chroma groups handed to the luma chains once the chroma chains are done (HM_CHAIN_LATE):
# kernel times at five waves per SIMD (17 spilled registers) and at four (no spill), then the GPU tests under the HM_CHAIN_LATE build
VARIANTS="-DHM_CHAIN_LATE=0|-DHM_CHAIN_LATE=1|-DHM_CHAIN_LATE=1 -DHM_WPE=4|-DHM_CHAIN_LATE=0 -DHM_WPE=4" OBJ=chain MODE=bench tools/probe_chain.sh
cd heif-decoder-lib_amd/csrc && rm -f build/hip_chain.o && make HIPFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I. -I/opt/rocm/include -DHM_CHAIN_LATE=1" >/dev/null 2>&1; cd ../..
echo "== GPU tests under -DHM_CHAIN_LATE=1 (without the zero-scratch test)"
timeout 1500 python3 -m pytest tests -m gpu -x -q --deselect tests/test_chain_modes_gpu.py::test_hot_path_kernels_hold_their_registers_without_a_spill 2>&1 | tail -3
echo "== bench parity gate + counters under -DHM_CHAIN_LATE=1"
python3 bench.py --quick --steps 5 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config'].get('parity'), {k: v['ms_per_step'] for k, v in d['kernels'].items()})"
cd heif-decoder-lib_amd/csrc && rm -f build/hip_chain.o && make >/dev/null 2>&1
