"""Experiment: where one GEMM workgroup's cycles go (needs a -DGSCAN_GEMM_STAMPS build, tools/variants.py).
    python tools/variants.py gst:all:-DGSCAN_GEMM_STAMPS,-DGSCAN_TRACE && GSCAN_HIP_LIB=variants/libgscan_hip.gst.so python tools/gemm_stamps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gpu_ops
from multimodal_seq2seq_gscan_amd import _lib
lib = _lib.load()
SHAPES = [("conv", 256, 5400, 576, "nn", 1), ("uv", 9216, 400, 150, "nt", 1), ("ge", 5120, 400, 100, "nt", 1),
          ("gx", 2560, 400, 25, "nt", 1), ("dS+=", 5120, 300, 500, "nn", 1), ("dW_ih s8", 400, 300, 5120, "tn", 8),
          ("dW_qt s8", 100, 100, 5120, "tn", 8), ("4096^3", 4096, 4096, 4096, "nt", 1),
          ("dW_ih s16", 400, 300, 5120, "tn", 16), ("dW_ih s24", 400, 300, 5120, "tn", 24),
          ("dW_hh s16", 400, 100, 5120, "tn", 16), ("dW_hh s32", 400, 100, 5120, "tn", 32),
          ("dW_qt s32", 100, 100, 5120, "tn", 32), ("de", 5120, 100, 400, "nn", 1), ("pkv", 9216, 100, 150, "nt", 1)]
NAMES = ["setup", "issue first", "first landed", "issue next (sum)", "reads+mfma (sum)", "wait+stage (sum)", "barrier (sum)", "epilogue"]
if os.environ.get("GSCAN_GEMM_MT", "1") != "0":    # gemm_mt.hip numbers its stamps differently
    NAMES = ["setup", "issue first", "wait+stage (sum)", "barrier A (sum)", "issue next (sum)", "reads+mfma (sum)", "barrier B (sum)", "epilogue(n/a)"]
for label, M, N, K, layout, split in SHAPES:
    A = torch.randn(M, K, device="cuda") if layout[0] == "n" else torch.randn(K, M, device="cuda").t()
    B = torch.randn(K, N, device="cuda") if layout[1] == "n" else torch.randn(N, K, device="cuda").t()
    Cm = torch.zeros(M, N, device="cuda")
    args = ((A, 0, A.stride(0), A.stride(1)), (B, 0, B.stride(0), B.stride(1)), (Cm, 0, N), M, N, K)
    kw = dict(beta=1.0, split_k=split) if split > 1 else {}
    for _ in range(3):
        gpu_ops.gemm(*args, **kw)
    torch.cuda.synchronize()
    buf = torch.zeros(2 + 6 * 256, dtype=torch.int64, device="cuda")
    _lib.check(lib.gscan_trace_set(buf.data_ptr()), "trace_set")
    reps = 5
    for _ in range(reps):
        gpu_ops.gemm(*args, **kw)
    torch.cuda.synchronize()
    _lib.check(lib.gscan_trace_set(None), "trace_set")
    t = buf.cpu().tolist()[1500:1508]
    rounds = -(-(-(-K // split) if split > 1 else K) // 32)
    print(f"{label:10s} rounds={rounds:3d}  " + "  ".join(f"{n}={v / reps:.0f}" for n, v in zip(NAMES, t)) + f"  total={sum(t) / reps:.0f} cycles")
