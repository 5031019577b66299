"""Experiment: where a workgroup of the persistent single-shot GEMM (csrc/gemm_shortk.hip) spends its cycles.
    python tools/variants.py gst:all:-DGSCAN_GEMM_STAMPS,-DGSCAN_TRACE
    GSCAN_HIP_LIB=variants/libgscan_hip.gst.so GSCAN_GEMM_SHORTK=2 python tools/shortk_stamps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gpu_ops
from multimodal_seq2seq_gscan_amd import _lib
lib = _lib.load()
SHAPES = [("uv", 9216, 400, 150), ("ge", 5120, 400, 100), ("4x uv", 36864, 400, 150)]
NAMES = ["first fetch", "wait+stage (sum)", "barrier (sum)", "fetch next (sum)", "reads+mfma (sum)", "barrier (sum)", "epilogue (sum)"]
for label, M, N, K in SHAPES:
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda").t()
    Cm = torch.zeros(M, N, device="cuda")
    args = ((A, 0, A.stride(0), A.stride(1)), (B, 0, B.stride(0), B.stride(1)), (Cm, 0, N), M, N, K)
    for _ in range(3):
        gpu_ops.gemm(*args)
    torch.cuda.synchronize()
    buf = torch.zeros(2 + 6 * 256, dtype=torch.int64, device="cuda")
    _lib.check(lib.gscan_trace_set(buf.data_ptr()), "trace_set")
    reps = 5
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        gpu_ops.gemm(*args)
    t1.record()
    torch.cuda.synchronize()
    _lib.check(lib.gscan_trace_set(None), "trace_set")
    t = buf.cpu().tolist()[1500:1507]
    tiles = -(-M // 64) * -(-N // 64)
    print(f"{label:8s} tiles={tiles:5d} ({tiles / 512:.1f} per workgroup)  " + "  ".join(f"{n}={v / reps:.0f}" for n, v in zip(NAMES, t))
          + f"  total={sum(t) / reps:.0f} cycles  launch={t0.elapsed_time(t1) / reps * 1e3:.1f} us")
