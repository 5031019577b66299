import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from multimodal_seq2seq_gscan_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "multimodal_seq2seq_gscan_amd", "libgscan_hip_stamps.so")
import gpu_ops
lib = _lib.load()
lib.gscan_debug_gemm_stamps.argtypes = [C.c_void_p, C.c_int]
def run(label, M, N, K, layout, split):
    A = torch.randn(M, K, device="cuda") if layout[0] == "n" else torch.randn(K, M, device="cuda").t()
    B = torch.randn(K, N, device="cuda") if layout[1] == "n" else torch.randn(N, K, device="cuda").t()
    Cm = torch.zeros(M, N, device="cuda")
    args = ((A, 0, A.stride(0), A.stride(1)), (B, 0, B.stride(0), B.stride(1)), (Cm, 0, N), M, N, K)
    kw = dict(beta=1.0, split_k=split) if split > 1 else {}
    gpu_ops.gemm(*args, **kw); torch.cuda.synchronize()
    lib.gscan_debug_gemm_stamps(None, 1)
    reps = 5
    for _ in range(reps): gpu_ops.gemm(*args, **kw)
    torch.cuda.synchronize()
    out = (C.c_longlong * 8)()
    lib.gscan_debug_gemm_stamps(out, 0)
    nt = -(-(-(-K // split)) // 32) if split > 1 else -(-K // 32)
    v = [x / reps for x in out]
    print(f"{label:10s} iters={nt:3d} prologue(load)={v[0]:7.0f} first store+bar={v[1]:6.0f} | per iter: issue={v[2]/nt:6.0f} compute={v[3]/nt:6.0f} store={v[4]/nt:6.0f} barrier={v[5]/nt:6.0f} | epilogue={v[6]:6.0f}  (100 MHz ticks?)")
for sh in [("dW_qt s8", 100, 100, 5120, "tn", 8), ("dW_ih s8", 400, 300, 5120, "tn", 8), ("dS+=", 5120, 300, 500, "nn", 1), ("uv", 9216, 400, 150, "nt", 1), ("ge", 5120, 400, 100, "nt", 1), ("conv", 256, 5400, 576, "nn", 1), ("4096^3", 4096, 4096, 4096, "nt", 1)]:
    run(*sh)
