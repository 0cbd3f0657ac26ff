"""Kernel timeline of one steady-state feedback round out of a rocprofv3 --kernel-trace csv:
   python tools/round_trace.py <dir with *_kernel_trace.csv> [first kernel substring = k_inc_seed] [which occurrence from the end = 3]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "k_inc_seed"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
i0 = starts[-back]
i1 = starts[-back + 1] if back > 1 else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = None
busy = 0
print(f"{path}: round starting at kernel #{i0}")
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("ssw::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    busy += e - s
    print(f"  +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f} us  gap {gap:6.1f} us  grid {r.get('Grid_Size', '?'):>9} wg {r.get('Workgroup_Size', '?'):>5}  {name[:70]}")
    prev_end = e
print(f"  span {(prev_end - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, {i1 - i0} launches")
