"""rocprofv3 --stats charges a kernel with the time it spends QUEUED behind another stream's kernel: the step launches
world_channel_lists_kernel on a side stream BEFORE the decoder's reverse recurrence, whose workgroups fill every CU's
register file, so the trace's start-to-end of that 5-7 us kernel is ~110 us and the table ranks it third.

    python tools/kernel_stats_fix.py <kernel_stats.csv of the --stats run> <kernel_trace.csv of a SERIALISED pass> > kernel_stats.csv

Rows of the kernels named in QUEUED get their durations from the second trace — a `--pmc` pass, in which rocprofv3 runs one
kernel at a time — and are marked in a new last column; percentages are recomputed over the corrected totals and the rows
re-sorted.  The uncorrected file is kept beside it as kernel_stats_raw.csv by tools/gpu_round.sh."""
import csv
import sys

QUEUED = ("world_channel_lists_kernel",)


def main():
    stats_path, trace_path = sys.argv[1], sys.argv[2]
    rows = [r for r in csv.DictReader(line for line in open(stats_path) if not line.startswith("#"))]
    alone = {}
    for r in csv.DictReader(open(trace_path)):
        for q in QUEUED:
            if q in r["Kernel_Name"]:
                alone.setdefault(q, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for r in rows:
        r["Source"] = "kernel trace of the --stats run"
        for q in QUEUED:
            if q in r["Name"] and alone.get(q):
                d = alone[q]
                avg = sum(d) / len(d)
                r["AverageNs"] = f"{avg:.6f}"
                r["TotalDurationNs"] = str(round(avg * int(r["Calls"])))
                r["MinNs"], r["MaxNs"] = str(min(d)), str(max(d))
                r["StdDev"] = f"{(sum((x - avg) ** 2 for x in d) / len(d)) ** 0.5:.6f}"
                r["Source"] = (f"durations from the serialised PMC pass ({len(d)} launches): the --stats trace counts ~100 us of "
                               "queueing behind decoder_bwd_kernel as this kernel's time")
    total = sum(int(r["TotalDurationNs"]) for r in rows) or 1
    for r in rows:
        r["Percentage"] = f"{100.0 * int(r['TotalDurationNs']) / total:.4g}"
    rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
    w = csv.DictWriter(sys.stdout, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
    w.writeheader()
    w.writerows(rows)


if __name__ == "__main__":
    main()
