"""SQ stall breakdown of two decode-GEMM shapes (development aid).
GPU box: rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/sq -o p -- python tools/pmc_sq.py run
         python tools/pmc_sq.py sum /tmp/sq"""
import csv, glob, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import torch
    from tools.bench_ops import time_gemm
    for (M, N, K) in [(64, 4096, 2048), (16, 4096, 2048), (64, 12288, 2048)]:
        time_gemm(M, N, K, cdt=torch.bfloat16, iters=50)
else:
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
    for path in glob.glob(f"{sys.argv[2]}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path, newline="")):
            if "gemm_skinny" not in row["Kernel_Name"]:
                continue
            key = row["Kernel_Name"].split("gemm_skinny_kernel")[1][:40] + " grid" + row.get("Grid_Size", "?")
            acc[key][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[key] += 1
    for k, d in acc.items():
        wc = d.get("SQ_WAVE_CYCLES", 1.0)
        print(k, {c: f"{v / wc:.3f}" for c, v in d.items()}, "wave_cycles/dispatch", wc / max(cnt[k] / max(len(d), 1), 1))
