"""Thin torch-tensor wrappers over the C ABI, used by the GPU tests only."""
from __future__ import annotations

import ctypes as C

import torch

from multimodal_seq2seq_gscan_amd import _lib


def stream():
    return torch.cuda.current_stream().cuda_stream


def gemm(a_view, b_view, c, M, N, K, *, alpha=1.0, beta=0.0, bias=None, act=0, mask=None, split_k=1):
    """a_view/b_view/c: (tensor, element offset, stride0, stride1) describing A(m,k), B(k,n) and C(row stride)."""
    lib = _lib.load()
    (a, ao, sam, sak), (b, bo, sbk, sbn), (ct, co, ldc) = a_view, b_view, c
    rc = lib.gscan_gemm_f32(M, N, K, alpha, a.data_ptr() + 4 * ao, sam, sak, b.data_ptr() + 4 * bo, sbk, sbn, beta,
                            ct.data_ptr() + 4 * co, ldc, _lib.ptr(bias), act,
                            None if mask is None else mask.data_ptr() + 4 * co, split_k, stream())
    _lib.check(rc, "gscan_gemm_f32")


def gemm_scratch(a_view, b_view, c, M, N, K, *, alpha=1.0, beta=0.0, bias=None, act=0, mask=None, split_k=1, asum=None,
                 scratch=None):
    """gscan_gemm_f32_scratch: the product as the training step issues it (row sums of A, split-K slabs in `scratch`)."""
    lib = _lib.load()
    (a, ao, sam, sak), (b, bo, sbk, sbn), (ct, co, ldc) = a_view, b_view, c
    rc = lib.gscan_gemm_f32_scratch(M, N, K, alpha, a.data_ptr() + 4 * ao, sam, sak, b.data_ptr() + 4 * bo, sbk, sbn, beta,
                                    ct.data_ptr() + 4 * co, ldc, _lib.ptr(bias), act,
                                    None if mask is None else mask.data_ptr() + 4 * co, split_k, _lib.ptr(asum),
                                    _lib.ptr(scratch), 0 if scratch is None else scratch.numel(), stream())
    _lib.check(rc, "gscan_gemm_f32_scratch")


def matmul(A: torch.Tensor, B: torch.Tensor, **kw) -> torch.Tensor:
    """Plain C = A @ B for 2-D tensors of any strides."""
    M, K = A.shape
    K2, N = B.shape
    assert K == K2
    Cm = torch.zeros(M, N, device=A.device)
    gemm((A, 0, A.stride(0), A.stride(1)), (B, 0, B.stride(0), B.stride(1)), (Cm, 0, N), M, N, K, **kw)
    return Cm
