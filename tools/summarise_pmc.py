"""Mean per dispatch of every counter rocprofv3 wrote under <dir>/*/ (csv), for kernels whose name contains <filter>,
plus the kernel-trace average duration.  usage: python tools/summarise_pmc.py <dir> <name filter>"""
import csv
import glob
import os
import sys
from collections import defaultdict

root, flt = sys.argv[1], sys.argv[2]


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


sums, counts = defaultdict(float), defaultdict(int)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            if flt not in name:
                continue
            key = (short(name), row["Counter_Name"])
            sums[key] += float(row["Counter_Value"])
            counts[key] += 1
dur, dcount = defaultdict(float), defaultdict(int)
for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            if flt not in name:
                continue
            k = short(name)
            dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            dcount[k] += 1
print("kernel,counter,mean_per_dispatch,dispatches")
for k in sorted(dur):
    print(f"{k},duration_us,{dur[k] / dcount[k] / 1e3:.3f},{dcount[k]}")
for (k, c) in sorted(sums):
    print(f"{k},{c},{sums[(k, c)] / counts[(k, c)]:.1f},{counts[(k, c)]}")
