"""Print the ordered kernel sequence (short name, grid, duration) of the LAST `n` launches of a rocprofv3 kernel trace CSV."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tot = 0
for r in rows[-n:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    name = r["Kernel_Name"].replace("void s2s::", "").replace("__hip_bfloat16", "bf16")[:70]
    print(f"{d:8.1f} us  grid {r['Grid_Size_X']:>8s}x{r['Grid_Size_Y']:<4s} wg {r['Workgroup_Size_X']:>4s}  {name}")
print(f"total {tot / 1e3:.2f} ms over {n} launches")
