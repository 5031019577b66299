"""Diagnostic: device time of gscan_gemm_f32 on the training step's product shapes, one launch at a time.

Run under rocprofv3 --kernel-trace and feed the CSV to this script with --parse: kernel durations come from the
trace (host launch overhead does not hide in them).

    rocprofv3 --kernel-trace --output-format csv -d out -o run -- python3 tools/gemm_shapes.py
    python tools/gemm_shapes.py --parse out/run_kernel_trace.csv
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SHAPES = [  # (label, M, N, K, layout, split)
    ("conv", 256, 5400, 576, "nn", 1), ("uv", 9216, 400, 150, "nt", 1), ("pkv", 9216, 100, 150, "nt", 1),
    ("ge", 5120, 400, 100, "nt", 1), ("gx", 2560, 400, 25, "nt", 1), ("ut", 2560, 400, 100, "nt", 1),
    ("pkt", 2560, 100, 100, "nt", 1), ("dS+=", 5120, 300, 500, "nn", 1), ("dS+= s2", 5120, 300, 500, "nn", 2),
    ("dS+= s4", 5120, 300, 500, "nn", 4),
    ("dW_ih s8", 400, 300, 5120, "tn", 8), ("dW_ih s16", 400, 300, 5120, "tn", 16),
    ("dW_hh s8", 400, 100, 5120, "tn", 8), ("dW_hh s16", 400, 100, 5120, "tn", 16), ("dW_hh s32", 400, 100, 5120, "tn", 32),
    ("dW_qt s8", 100, 100, 5120, "tn", 8), ("dW_qt s32", 100, 100, 5120, "tn", 32),
    ("dW_kv s15", 100, 150, 9216, "tn", 15), ("dW_kv s58", 100, 150, 9216, "tn", 58),
    ("dWt", 576, 5400, 256, "tn", 1), ("dW_enc s8", 400, 100, 2560, "tn", 8), ("dW_enc s16", 400, 100, 2560, "tn", 16),
    ("dxe s8", 2560, 25, 800, "nn", 8), ("4096^3", 4096, 4096, 4096, "nt", 1),
    ("uv K152", 9216, 400, 152, "nt", 1), ("uv K160", 9216, 400, 160, "nt", 1), ("uv K128", 9216, 400, 128, "nt", 1),
    ("uv N384", 9216, 384, 160, "nt", 1), ("ge K128", 5120, 400, 128, "nt", 1),
    ("dW_ih s8 @32", 400, 300, 5120, "tn", 8), ("dW_hh s8 @32", 400, 100, 5120, "tn", 8), ("dW_qt s8 @32", 100, 100, 5120, "tn", 8),
    ("dS+= @32", 5120, 300, 500, "nn", 1), ("uv @32", 9216, 400, 150, "nt", 1), ("ge @32", 5120, 400, 100, "nt", 1),
    ("de", 5120, 100, 400, "nn", 1), ("de @32", 5120, 100, 400, "nn", 1),
]
REPS = 3


def run():
    import torch
    import gpu_ops
    for label, M, N, K, layout, split in SHAPES:
        pad = (lambda n: (n + 31) // 32 * 32) if label.endswith("@32") else (lambda n: n)    # row pitch padded to 128 bytes
        A = torch.randn(M, pad(K), device="cuda")[:, :K] if layout[0] == "n" else torch.randn(K, pad(M), device="cuda")[:, :M].t()
        B = torch.randn(K, pad(N), device="cuda")[:, :N] if layout[1] == "n" else torch.randn(N, pad(K), device="cuda")[:, :K].t()
        Cm = torch.zeros(M, N, device="cuda")
        args = ((A, 0, A.stride(0), A.stride(1)), (B, 0, B.stride(0), B.stride(1)), (Cm, 0, N), M, N, K)
        kw = dict(beta=1.0, split_k=split) if split > 1 else {}
        if os.environ.get("GSCAN_SHAPES_SCRATCH") == "1":      # split-K through slabs (gemm_mt.hip) instead of atomics
            if "scratch" not in globals():
                globals()["scratch"] = torch.empty(48 << 20, device="cuda")
            kw["scratch"] = globals()["scratch"]
            for _ in range(REPS):
                gpu_ops.gemm_scratch(*args, **kw)
            continue
        for _ in range(REPS):
            gpu_ops.gemm(*args, **kw)
        torch.cuda.synchronize()


def parse(path):
    import csv
    allrows = [r for r in csv.DictReader(open(path)) if any(k in r["Kernel_Name"] for k in ("gemm_group_kernel", "gemm_wide_kernel", "gemm_mt_kernel", "gemm_mt_reduce_kernel"))]
    allrows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = []          # [main duration, reduce duration, kernel, grid]
    for r in allrows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if "gemm_mt_reduce_kernel" in r["Kernel_Name"]:
            rows[-1][1] += d
        else:
            rows.append([d, 0, r["Kernel_Name"].split("(")[0].split("::")[-1][:18], int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(1, int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1))))])
    assert len(rows) == REPS * len(SHAPES), (len(rows), len(SHAPES))
    for i, (label, M, N, K, layout, split) in enumerate(SHAPES):
        grp = sorted(rows[REPS * i:REPS * (i + 1)], key=lambda x: x[0] + x[1])
        main, red, name, wgs = grp[len(grp) // 2]
        us = (main + red) / 1e3
        print(f"{label:12s} {M:5d}x{N:5d}x{K:5d} {layout} split={split:2d}: {us:7.1f} us (reduce {red / 1e3:4.1f})  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s  {name} x{wgs}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        run()
