#!/bin/bash
# Instruction counts (SQ_INSTS_*) of one kernel for A/B builds of its source file (see tools/probe_chain.sh for VARIANTS),
# 48 images.  OBJ=chain|filters|residual (the object rebuilt), KERNEL=the kernel name to sum.
export TMPDIR=/tmp
OBJ=${OBJ:-chain}; export KERNEL=${KERNEL:-k_chain}
cd heif-decoder-lib_amd/csrc
IFS='|' read -ra VS <<< "${VARIANTS:--DHM_Q_PROBE=0|-DHM_Q_PROBE=1|-DHM_Q_PROBE=2|-DHM_Q_PROBE=3}"
for v in "${VS[@]}"; do
  rm -f build/hip_$OBJ.o
  make HIPFLAGS="--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -I../../include -I. -I/opt/rocm/include $v" >/dev/null 2>&1
  out=/tmp/pmcv; rm -rf $out; mkdir -p $out
  (cd ../.. && rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out --output-format csv -- python3 bench.py --no-parity --quick --steps 2 --warmup 1 --images 48 > $out/log 2>&1)
  echo -n "variant [$v]: "
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
done
rm -f build/hip_$OBJ.o; make >/dev/null 2>&1
