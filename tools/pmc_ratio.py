"""FETCH_SIZE / WRITE_SIZE of the decode-GEMM microbenchmark (tools/bench_ops.py skinny) against its algorithmic bytes.
usage (GPU box): rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p -o p -- python tools/bench_ops.py skinny ; python tools/pmc_ratio.py /tmp/p out.json"""
import csv, glob, json, sys
from collections import defaultdict

SHAPES = [(64, 4096, 1024), (64, 1024, 2048), (64, 6144, 1024), (64, 1024, 3072), (64, 4096, 2048), (64, 2048, 2048),
          (64, 12288, 2048), (64, 2048, 6144), (64, 3840, 1280), (64, 1280, 1280), (64, 5120, 1280), (64, 1280, 5120),
          (64, 51866, 1280), (1, 4096, 2048), (16, 4096, 2048), (128, 4096, 1024)]
LAUNCHES = 205
alg = sum(LAUNCHES * (N * K * 2 + M * K * 2 + M * N * 2) for M, N, K in SHAPES)
tot = defaultdict(float)
n = defaultdict(int)
for path in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path, newline="")):
        if "gemm_skinny_kernel" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            n[row["Counter_Name"]] += 1
res = {"algorithmic_bytes": alg, "dispatches": dict(n), "counter_sums": dict(tot)}
json.dump(res, open(sys.argv[2], "w"), indent=1)
print(res)
