#!/bin/bash
# HBM traffic of the bench kernels from the TCC counters, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (--kernel-trace only), bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024
# (gfx950: FETCH_SIZE counts 64 B per 128-B request).  Writes profiles-style JSON to gpurun_out/pmc_traffic_<tag>.json.
# usage (GPU box, repo root): tools/pmc_traffic.sh <tag> <images>
tag=$1; images=${2:-16}
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_traffic_$tag
mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $out/$c --output-format csv -- python3 bench.py --no-parity --quick --steps 3 --warmup 1 --images $images > $out/$c.log 2>&1
  echo "$c rc=$?"
done
python3 - "$out" "$images" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
root, images = sys.argv[1], int(sys.argv[2])
names = {"k_recon": "k_recon_quad", "k_chain": "k_chain", "k_residual": "k_residual", "k_deblock": "k_deblock", "k_sao_paste": "k_sao_paste", "k_ycbcr420_int": "k_ycbcr420_int(colour)", "k_tail420": "k_tail420(deblock+sao+paste+colour)"}
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            for k in names:
                if k in row["Kernel_Name"] and row["Counter_Name"] == c:
                    a = acc[k][c]; a[0] += float(row["Counter_Value"]); a[1] += 1
res = {"source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), python3 bench.py --steps 3 --warmup 1 --images {images}, MI355X; values are averages per launch in KiB as reported (tools/pmc_traffic.sh)",
       "correction": "MI355X_MICROARCH.md HBM section: bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE counts 64 B per 128-B request on gfx950; calibrated for wide coalesced streams - narrow accesses are an upper bound)",
       "images_per_launch": images, "kernels": {}}
launches = {"k_recon": 1, "k_chain": 1, "k_residual": 1, "k_deblock": 1, "k_sao_paste": 1, "k_tail420": 1, "k_ycbcr420_int": (images + 31) // 32}  # hm_colour_convert_batch: 32 images per launch
for k, n in names.items():
    if not acc[k]["FETCH_SIZE"][1]:
        continue
    f = acc[k]["FETCH_SIZE"][0] / acc[k]["FETCH_SIZE"][1]
    w = acc[k]["WRITE_SIZE"][0] / max(1, acc[k]["WRITE_SIZE"][1])
    per_step = (2 * f + w) * 1024 * launches[k]
    res["kernels"][n] = {"FETCH_SIZE_KiB_per_launch": round(f, 1), "WRITE_SIZE_KiB_per_launch": round(w, 1),
                         "launches_per_step": launches[k], "hbm_bytes_per_step": int(per_step), "hbm_bytes_per_image": int(per_step / images)}
json.dump(res, open(root + ".json", "w"), indent=1)
print(json.dumps(res["kernels"], indent=1))
PY
