"""Kernel families of a rocprofv3 --kernel-trace --stats summary (the *_kernel_stats.csv), per turn-batch:
usage: python tools/prof_families.py profiles/r02_c3_serial_kernel_stats.csv <n_turn_batches> [bench_line.json]
With the bench line of the same command, recomputes every roofline fraction from the CSV durations and the line's algorithmic bytes / flops."""
import csv
import json
import sys

FAMILIES = {"gemm_dec": ("gemm_dec_kernel", "gemm_dec2_kernel", "gemm_decw_kernel", "gemm_skinny_kernel"), "gemm_tile": ("gemm_tile3_kernel", "gemm_tile4_kernel", "gemm_tile5_kernel", "gemm_tile_kernel", "conv7_tile_kernel"), "attn_mfma": ("attn_mfma_kernel",),
            "attn_rows": ("attn_step_kernel", "attn_step_small_kernel", "attn_decode_kernel", "attn_rows_kernel")}
rows = list(csv.DictReader(open(sys.argv[1])))
nb = int(sys.argv[2])
line = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
total = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"all kernels: {total / nb / 1e6:.1f} ms per turn-batch, {sum(int(r['Calls']) for r in rows) // nb} launches")
for fam, pats in FAMILIES.items():
    sel = [r for r in rows if any(p in r["Name"] for p in pats)]
    ns, calls = sum(float(r["TotalDurationNs"]) for r in sel), sum(int(r["Calls"]) for r in sel)
    if not calls:
        continue
    msg = f"{fam:10s} {ns / nb / 1e6:8.2f} ms per turn-batch  {calls // nb:7d} launches  avg {ns / calls / 1e3:7.2f} us"
    if line and line.get("roofline") and fam in line["roofline"].get("families", {}):
        f = line["roofline"]["families"][fam]
        full = line["roofline"] if line["roofline"].get("kernel_ms_per_step") == f["kernel_ms_per_step"] else None
        msg += f"   | bench line: {f['kernel_ms_per_step']} ms, avg {f['avg_kernel_us']} us, frac {f['frac']}"
    print(msg)
