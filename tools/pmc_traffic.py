"""HBM-side traffic of the dominant kernel family from two rocprofv3 PMC passes of bench.py.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -o run -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -o run -- python3 bench.py ...
    python tools/pmc_traffic.py out_f/run_counter_collection.csv out_w/run_counter_collection.csv \
        gemm_group_kernel,decoder_fwd_kernel,decoder_bwd_kernel,keys_backward_kernel

Prints one JSON object: average bytes per launch of every named kernel (the first one also at the top level).  FETCH_SIZE / WRITE_SIZE are in KiB-like units
of 1024 bytes... the raw unit is kilobytes; on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane)
coalesced reads (MI355X_MICROARCH.md, HBM section), so it is doubled here; WRITE_SIZE is exact for 16 B/lane
stores and float atomics.
"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def total(path, counter, kernel):
    s, n = 0.0, 0
    ids = set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
            s += float(r["Counter_Value"])
            ids.add(r["Dispatch_Id"])
    return s, len(ids)


def main():
    fpath, wpath, kernels = sys.argv[1], sys.argv[2], sys.argv[3].split(",")
    per = {}
    for kernel in kernels:
        f, nf = total(fpath, "FETCH_SIZE", kernel)
        w, nw = total(wpath, "WRITE_SIZE", kernel)
        per[kernel] = {"launches_fetch_pass": nf, "launches_write_pass": nw,
                       "fetch_bytes_per_launch": 2.0 * 1024.0 * f / max(nf, 1),
                       "write_bytes_per_launch": 1024.0 * w / max(nw, 1)}
        per[kernel]["traffic_bytes_per_launch"] = per[kernel]["fetch_bytes_per_launch"] + per[kernel]["write_bytes_per_launch"]
    first = kernels[0]            # the family bench.py's roofline block prices (its keys stay at the top level)
    out = {"kernel": first, **per[first],
           "corrections": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), unit 1 KB; WRITE_SIZE x1",
           "kernels": per}
    from bench import source_hash          # the profile is valid for exactly these kernel sources (bench.py checks)
    out["source_sha"] = source_hash()
    out["tag"] = sys.argv[4] if len(sys.argv) > 4 else ""
    out["workload"] = "compositional"
    print(json.dumps(out))


if __name__ == "__main__":
    main()
