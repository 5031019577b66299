"""Timeline of one training step from a rocprofv3 --kernel-trace CSV.

    python tools/timeline.py <kernel_trace.csv> [step_index_from_end]

Prints every kernel of one step (delimited by the Adam kernel): start offset in the step, duration, HSA queue
(= stream) and name, so that the critical chain and the overlap between the two streams can be read off.
"""
import csv
import sys


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("gscan::adam") or "adam_kernel" in r["Kernel_Name"]]
    if len(ends) < back + 1:
        raise SystemExit("not enough steps in the trace")
    lo, hi = ends[-back - 1] + 1, ends[-back]
    t0 = int(rows[lo]["Start_Timestamp"])
    queues = {}
    for r in rows[lo:hi + 1]:
        q = queues.setdefault(r["Queue_Id"], len(queues))
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        name = r["Kernel_Name"].split("(")[0].replace("gscan::", "")[:60]
        print(f"{s / 1e3:8.1f} {(e - s) / 1e3:7.1f}  q{q}  {'    ' * q}{name}  grid={r.get('Grid_Size', '')}")
    print(f"span {(int(rows[hi]['End_Timestamp']) - t0) / 1e3:.1f} us, {hi - lo + 1} kernels")


if __name__ == "__main__":
    main()
