#!/usr/bin/env python3
"""Average the rocprofv3 counter_collection.csv values per kernel and counter (per launch)."""
import csv, glob, json, os, sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"]
            for k in ("k_recon", "k_chain", "k_residual", "k_deblock", "k_sao_paste", "k_ycbcr", "k_tail420"):
                if k in name:
                    a = acc[k][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"]); a[1] += 1
out = {k: {c: v[0] / v[1] for c, v in sorted(cs.items())} for k, cs in acc.items()}
for k, cs in out.items():
    if cs.get("SQ_THREAD_CYCLES_VALU") and cs.get("SQ_ACTIVE_INST_VALU"):
        # thread-cycles over wave-cycles of vector instructions: average active lanes (of 64)
        cs["_valu_active_lanes_of_64"] = round(cs["SQ_THREAD_CYCLES_VALU"] / cs["SQ_ACTIVE_INST_VALU"], 2)
    wc = cs.get("SQ_WAVE_CYCLES")
    if wc:
        cs["_frac_of_wave_cycles"] = {c: round(cs[c] / wc, 4) for c in cs if c.startswith(("SQ_WAIT", "SQ_ACTIVE")) }
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
for k in ("k_recon", "k_chain", "k_residual"):
    if k in out:
        print(k, json.dumps(out[k], indent=1))
