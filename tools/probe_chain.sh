#!/bin/bash
# A/B builds of one kernel source on the GPU box.  Every entry of VARIANTS ('|'-separated) is a set of extra compiler flags
# ("-DHM_Q_PROBE=n" compiles parts of k_chain's loop out - pictures wrong, parity gate off -: 1 no 4x4 path, 2 no wave-wide
# path, 4 wave-wide path without prediction, 512 no residual of the large blocks, 1024 the CTU flush without its stores to the picture, 2048 no residual loads of the
# 8x8 blocks, 4096 windows of micro-ops never reloaded; "-DHM_WPE=n" sets the waves per SIMD the
# register allocation aims for; "-DHM_T_PROBE=n" the same idea in filters.hip).
#   OBJ=chain|filters|residual   the object rebuilt per variant (default chain)
#   MODE=bench|few|tiles|counters  bench: kernel times of `bench.py --quick` (default); few: tools/few_pictures_probe.py
#                                (32 large pictures per class); tiles: both reconstruction kernels for TILES="256 1024" 512x512 tiles
#                                (tools/bench_classes.py, CLASS=8bit_420_ctb32); counters: SQ instruction counts of KERNEL per tile
#                                (rocprofv3 --pmc, 48 images; PMC="..." replaces the counter list)
# usage (repo root): VARIANTS="-DHM_Q_PROBE=1|-DHM_Q_PROBE=2" [OBJ=..] [MODE=..] tools/probe_chain.sh [bench args]
OBJ=${OBJ:-chain}; MODE=${MODE:-bench}; export KERNEL=${KERNEL:-k_chain}
export TMPDIR=/tmp
cd heif-decoder-lib_amd/csrc
IFS='|' read -ra VS <<< "${VARIANTS:--DHM_Q_PROBE=1|-DHM_Q_PROBE=2|-DHM_Q_PROBE=3}"
for v in "${VS[@]}"; do
  rm -f build/hip_$OBJ.o
  make HIPFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I. -I/opt/rocm/include $v" >/dev/null 2>&1 || echo "BUILD FAILED [$v]"
  echo -n "variant [$v]: "
  case $MODE in
  bench)
    (cd ../.. && HM_CHAIN_DEBUG=1 python3 bench.py --quick --no-parity --steps 5 "$@" 2>/tmp/probe_err.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], {k: v['ms_per_step'] for k, v in d['kernels'].items()})"; grep -m1 "k_chain" /tmp/probe_err.log || true) ;;
  few)
    (cd ../.. && python3 tools/few_pictures_probe.py 2>/dev/null | tail -1) ;;
  tiles)
    (cd ../.. && for n in ${TILES:-256 1024}; do echo -n "$n tiles: "; HM_CLASS_TILES=$n HM_CLASS_ONLY=${CLASS:-8bit_420_ctb32} python3 tools/bench_classes.py 2>/dev/null | tr -d '\n' | sed -E 's/.*k_recon_ms": ([0-9.]+).*/\1 ms  /'; done; echo) ;;
  counters)
    out=/tmp/pmcv; rm -rf $out; mkdir -p $out
    (cd ../.. && rocprofv3 --kernel-trace --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE} -d $out --output-format csv -- python3 bench.py --no-parity --quick --steps 2 --warmup 1 --images 48 > $out/log 2>&1)
    python3 - $out <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if os.environ["KERNEL"] in row["Kernel_Name"]:
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
print({k: round(v[0] / v[1] / 2304 / 1000, 1) for k, v in sorted(acc.items())}, "(thousands per tile)")
PY
    ;;
  esac
done
rm -f build/hip_$OBJ.o; make >/dev/null 2>&1
