"""Per-dispatch counter values of a rocprofv3 --pmc run for kernels whose name contains a pattern:
    python tools/pmc_dump.py <dir> <pattern> [labels ...]      (labels name the dispatches in launch order, e.g. the shapes of tools/tile5_pmc.py)"""
import csv
import glob
import sys
from collections import OrderedDict

d, pat, labels = sys.argv[1], sys.argv[2], sys.argv[3:]
rows = OrderedDict()
for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        if pat in r["Kernel_Name"]:
            rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for i, (did, c) in enumerate(sorted(rows.items())):
    name = labels[i] if i < len(labels) else f"dispatch {did}"
    extra = ""
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (it equals 16 cycles x the launch's v_mfma_f32_16x16x32_bf16 count exactly);
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs (/ 8 = the launch's duration in shader cycles: 1.6 M for an 850 us launch at 1.9 GHz)
        act = c["GRBM_GUI_ACTIVE"] / 8.0
        extra = f"   -> {act / 1e6:.2f} M active cycles per XCD, MFMA pipe busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (act * 1024):.3f} of them"
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
        extra += f"   LDS bank-conflict cycles {c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.3f} of the LDS-active cycles"
    print(f"{name:10s} " + "  ".join(f"{k} {v:.4g}" for k, v in sorted(c.items())) + extra)
