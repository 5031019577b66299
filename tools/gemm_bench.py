"""Diagnostic: achieved TFLOP/s of gscan_gemm_f32 on the step's product shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import gpu_ops

def run(M, N, K, layout, split=1, reps=20):
    A = torch.randn(M, K, device="cuda") if layout[0] == "n" else torch.randn(K, M, device="cuda").t()
    B = torch.randn(K, N, device="cuda") if layout[1] == "n" else torch.randn(N, K, device="cuda").t()
    C = torch.zeros(M, N, device="cuda")
    args = ((A, 0, A.stride(0), A.stride(1)), (B, 0, B.stride(0), B.stride(1)), (C, 0, N), M, N, K)
    kw = dict(beta=1.0, split_k=split) if split > 1 else {}
    for _ in range(3):
        gpu_ops.gemm(*args, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gpu_ops.gemm(*args, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"M={M:5d} N={N:5d} K={K:5d} {layout} split={split:2d}: {us:7.1f} us  {2.0*M*N*K/us/1e6:6.1f} TFLOP/s  tiles={((M+63)//64)*((N+63)//64)*split}")

for shape in [(5120, 400, 400, "nn", 1), (5120, 400, 100, "nt", 1), (5120, 100, 400, "nt", 1), (9216, 400, 150, "nt", 1),
              (9216, 50, 784, "nt", 1), (9216, 150, 784, "nt", 1), (400, 300, 5120, "tn", 8), (400, 300, 5120, "tn", 16),
              (400, 300, 5120, "tn", 40), (100, 100, 5120, "tn", 8), (100, 100, 5120, "tn", 40), (50, 784, 9216, "tn", 15),
              (4096, 4096, 4096, "nt", 1), (4096, 4096, 4096, "nn", 1), (256, 5400, 576, "nn", 1), (576, 5400, 256, "tn", 1)]:
    run(*shape)
