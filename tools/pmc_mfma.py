"""Matrix-core utilisation per kernel from one rocprofv3 PMC pass.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
              -d out -o run -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0
    python tools/pmc_mfma.py out/run_counter_collection.csv

Per kernel family: launches, MFMA-busy cycles per launch (SQ_VALU_MFMA_BUSY_CYCLES: cycles in which a SIMD's matrix
pipe is busy, summed over the chip's 1024 SIMDs), GPU-active cycles per launch (GRBM_GUI_ACTIVE is summed over the 8
XCDs, MI355X_MICROARCH.md DVFS section: divided by 8 here) and the ratio
    mfma_util = MFMA_BUSY / (GUI_ACTIVE/8 * 256 CUs * 4 SIMDs),
the fraction of the matrix pipes' cycles at the clock the chip actually held.  A v_mfma_f32_16x16x4_f32 keeps its
pipe busy 32 cycles for 2048 flops, so mfma_util is also the fraction of the fp32 matrix peak spent on issued
(padded) MFMAs; the bench line's roofline.frac counts algorithmic flops against the 2.4 GHz peak instead.
"""
import csv
import json
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    per = defaultdict(lambda: defaultdict(float))
    ids = defaultdict(set)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("gscan::", "")
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
        ids[name].add(r["Dispatch_Id"])
    out = {}
    for name, c in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
        n = len(ids[name])
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        out[name] = {"launches": n, "mfma_busy_cycles_per_launch": busy / n, "gpu_active_cycles_per_launch": gui / n,
                     "cu_busy_cycles_per_launch": c.get("SQ_BUSY_CU_CYCLES", 0.0) / n,
                     "mfma_util": busy / (gui * 1024.0) if gui > 0 else None}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
