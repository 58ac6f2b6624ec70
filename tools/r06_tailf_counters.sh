#!/bin/bash
# r06: SQ and TCC counters of k_tailf on the class of HDR photographs (tools/tailf_probe.py: 96 x 12 MP grids of 10-bit 4:2:0
# tiles -> RGB24 = 4608 tiles per launch), one rocprofv3 pass per counter group, FETCH_SIZE / WRITE_SIZE in passes of their own
export TMPDIR=/tmp
out=/tmp/pmc_tailf; rm -rf $out; mkdir -p $out
i=0
for g in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $g -d $out/g$i --output-format csv -- python3 $OLDPWD/tools/tailf_probe.py 3 > $out/g$i.log 2>&1)
done
python3 - $out <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: [0.0, 0])
names = set()
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_tailf" in row["Kernel_Name"] or "k_tail420" in row["Kernel_Name"]:  # (the HDR class: k_tail420<16-bit> since r06; HM_TAIL_HDR16=0: k_tailf)
            names.add(row["Kernel_Name"].split("(")[0][:80])
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
per = {k: v[0] / v[1] for k, v in acc.items()}
tiles = 4608
print("the fused tail of the HDR class (10-bit 4:2:0 -> RGB24), 4608 tiles of 512 x 512 per launch (deblocking + SAO + paste + depth change + integer matrix), averages over the probe's four launches; kernel:", sorted(names))
for k in sorted(per):
    print(f"  {k:24s} {per[k]:16.0f} per launch   {per[k] / tiles / 1000:10.2f} k per tile")
if per.get("SQ_THREAD_CYCLES_VALU") and per.get("SQ_ACTIVE_INST_VALU"):
    print("  active lanes per vector instruction: %.1f of 64" % (per["SQ_THREAD_CYCLES_VALU"] / per["SQ_ACTIVE_INST_VALU"]))
if per.get("FETCH_SIZE") and per.get("WRITE_SIZE"):
    b = (2 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024
    px = tiles * 512 * 512
    print("  HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, KiB -> B): %.2f GB = %.2f B per pixel (algorithmic: 3 B/px of samples in + 3 B/px of pixels out)" % (b / 1e9, b / px))
PY
