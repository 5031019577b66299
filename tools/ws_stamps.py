"""Experiment: where a workgroup of the weights-stationary GEMM (gemm_ws.hip) spends its cycles; needs a stamped build:
    python tools/variants.py wst:gemm_ws.hip:-DGSCAN_GEMM_STAMPS,-DGSCAN_TRACE
    GSCAN_GEMM_WS=2 GSCAN_HIP_LIB=variants/libgscan_hip.wst.so python tools/ws_stamps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gpu_ops
from multimodal_seq2seq_gscan_amd import _lib
lib = _lib.load()
SHAPES = [("uv", 9216, 400, 150), ("pkv", 9216, 100, 150), ("ge", 5120, 400, 100), ("ut", 2560, 400, 100), ("big", 36864, 400, 150)]
NAMES = ["B block", "wait copy (sum)", "barrier (sum)", "issue next (sum)", "reads+mfma (sum)", "stores (sum)", "tiles"]
for label, M, N, K in SHAPES:
    A, B, Cm = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.zeros(M, N, device="cuda")
    args = ((A, 0, K, 1), (B, 0, 1, K), (Cm, 0, N), M, N, K)
    for _ in range(3):
        gpu_ops.gemm(*args)
    torch.cuda.synchronize()
    buf = torch.zeros(2 + 6 * 256, dtype=torch.int64, device="cuda")
    _lib.check(lib.gscan_trace_set(buf.data_ptr()), "trace_set")
    reps = 5
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        gpu_ops.gemm(*args)
    ev1.record()
    torch.cuda.synchronize()
    _lib.check(lib.gscan_trace_set(None), "trace_set")
    t = buf.cpu().tolist()[1500:1507]
    print(f"{label:6s} {M}x{N}x{K}  {ev0.elapsed_time(ev1) / reps * 1e3:7.1f} us per launch | " +
          "  ".join(f"{n}={v / reps:.0f}" for n, v in zip(NAMES, t)) + f"  total={sum(t[:6]) / reps:.0f} cycles")
