"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel family (run on the GPU box; the raw CSV is too large to
travel).  usage: pmc_summary.py <dir> <out.json>"""
import csv, glob, json, re, sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for path in glob.glob(f"{src}/**/*counter_collection.csv", recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row["Kernel_Name"]
            m = re.search(r"(gemm_tile5_kernel|gemm_tile3_kernel|gemm_tile_kernel|gemm_skinny_kernel|attn_mfma_kernel|attn_decode_kernel|attn_rows_kernel|norm_rows_kernel|snake_kernel)", name)
            fam = m.group(1) if m else "other"
            a = acc[fam][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
res = {fam: {c: {"sum": v[0], "dispatches": v[1], "per_dispatch": v[0] / max(v[1], 1)} for c, v in d.items()} for fam, d in acc.items()}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
