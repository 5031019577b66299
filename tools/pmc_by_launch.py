"""Per-LAUNCH-SHAPE averages of PMC counters (rocprofv3 --pmc ... --kernel-trace --output-format csv) for one kernel:
launches are grouped by grid size, which tells the step's five GEMM launches apart.
    python tools/pmc_by_launch.py <counter_collection.csv> [kernel substring]"""
import csv
import sys
from collections import defaultdict

path, kernel = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "gemm_group_kernel")
acc = defaultdict(lambda: defaultdict(float))
ids = defaultdict(set)
for r in csv.DictReader(open(path)):
    if kernel not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"].split("(")[0][-28:], int(r["Grid_Size"]) // max(int(r.get("Workgroup_Size", "256") or 256), 1))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    ids[key].add(r["Dispatch_Id"])
names = sorted({c for v in acc.values() for c in v})
print("kernel / workgroups".ljust(40), "n".rjust(4), " ".join(n[-18:].rjust(18) for n in names))
for key in sorted(acc, key=lambda k: k[1]):
    n = len(ids[key])
    print(f"{key[0]} x{key[1]}".ljust(40), str(n).rjust(4), " ".join(f"{acc[key][c] / n:18.0f}" for c in names))
