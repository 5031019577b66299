"""HBM-side traffic of EVERY kernel of the training step from two rocprofv3 PMC passes of bench.py.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -o run -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -o run -- python3 bench.py ...
    python tools/pmc_traffic.py out_f/run_counter_collection.csv out_w/run_counter_collection.csv [ignored] [tag]

Prints one JSON object: average bytes per launch of every library kernel found in the two passes (names cut at the
template / argument list), each with its LAUNCHES PER STEP taken from the same trace — launches of the kernel divided by
launches of the optimiser kernel, which runs exactly once per training step — so that bench.py can add up a step without
assuming a schedule.  The grouped-GEMM family's figure also stays at the top level (bench.py's `roofline_gemm.traffic`).
FETCH_SIZE / WRITE_SIZE are in units of 1024 bytes; on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane)
coalesced reads (MI355X_MICROARCH.md, HBM section), so it is doubled here; WRITE_SIZE is exact for 16 B/lane stores and
float atomics.
"""
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STEP_MARKERS = ("adam_masks_kernel", "adam_kernel")      # one launch per training step


def short_name(full: str):
    """gscan::decoder_fwd_kernel<100, true, ...>(gscan::DecoderArgs) -> decoder_fwd_kernel; None for other libraries' kernels."""
    m = re.search(r"gscan::([A-Za-z_0-9]+)", full)
    return m.group(1) if m else None


def totals(path, counter):
    """{kernel: (sum of the counter, launches)} over the library's kernels."""
    s, ids = {}, {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short_name(r["Kernel_Name"])
        if k is None:
            continue
        s[k] = s.get(k, 0.0) + float(r["Counter_Value"])
        ids.setdefault(k, set()).add(r["Dispatch_Id"])
    return {k: (s[k], len(ids[k])) for k in s}


def main():
    fpath, wpath = sys.argv[1], sys.argv[2]
    tag = sys.argv[4] if len(sys.argv) > 4 else (sys.argv[3] if len(sys.argv) > 3 and "," not in sys.argv[3] else "")
    fetch, write = totals(fpath, "FETCH_SIZE"), totals(wpath, "WRITE_SIZE")
    steps_f = sum(fetch.get(m, (0, 0))[1] for m in STEP_MARKERS)
    steps_w = sum(write.get(m, (0, 0))[1] for m in STEP_MARKERS)
    per = {}
    for kernel in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(kernel, (0.0, 0))
        w, nw = write.get(kernel, (0.0, 0))
        e = {"launches_fetch_pass": nf, "launches_write_pass": nw,
             "fetch_bytes_per_launch": 2.0 * 1024.0 * f / max(nf, 1),
             "write_bytes_per_launch": 1024.0 * w / max(nw, 1),
             # from the trace itself, not from a schedule someone assumed (None: no optimiser launches in the pass)
             "launches_per_step": round(nf / steps_f, 3) if steps_f else None,
             "launches_per_step_write_pass": round(nw / steps_w, 3) if steps_w else None}
        e["traffic_bytes_per_launch"] = e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"]
        per[kernel] = e
    if not per:      # a failed run, or kernel names that no longer demangle to gscan::<name>: say so instead of StopIteration
        sys.exit(f"pmc_traffic: no gscan:: kernel with FETCH_SIZE / WRITE_SIZE rows in {fpath} / {wpath} "
                 "(did the profiled run fail, or has the name mangling changed?)")
    if steps_f != steps_w:
        print(f"pmc_traffic: warning: {steps_f} steps in the fetch pass, {steps_w} in the write pass — per-launch figures are "
              "per pass; launches_per_step is given for each pass", file=sys.stderr)
    first = "gemm_group_kernel" if "gemm_group_kernel" in per else next(iter(per))
    out = {"kernel": first, **per[first],
           "corrections": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), unit 1 KB; WRITE_SIZE x1",
           "steps_in_fetch_pass": steps_f, "steps_in_write_pass": steps_w,
           "kernels": per}
    from bench import source_hash          # the profile is valid for exactly these kernel sources (bench.py checks)
    out["source_sha"] = source_hash()
    out["tag"] = tag
    out["workload"] = "compositional"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
