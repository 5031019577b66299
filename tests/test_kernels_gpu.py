"""Each HIP kernel family against a plain PyTorch (CPU, fp32/fp64) statement of the same operation.
Called through the C ABI of libgscan_hip.so.  Run on the GPU box: pytest -m gpu."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from multimodal_seq2seq_gscan_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return _lib.load()


def dev(x):
    return x.to("cuda")


@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (9216, 150, 100), (37, 9, 100), (400, 300, 5120), (5, 7, 3),
                                   (130, 70, 33)])
@pytest.mark.parametrize("layout", ["nn", "nt", "tn"])
def test_gemm_layouts(lib, M, N, K, layout):
    import gpu_ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    ref = (A.double() @ B.double())
    Ad = dev(A) if layout != "tn" else dev(A.t().contiguous()).t()       # tn: A stored [K,M]
    Bd = dev(B) if layout != "nt" else dev(B.t().contiguous()).t()       # nt: B stored [N,K]
    out = gpu_ops.matmul(Ad, Bd).cpu()
    err = (out.double() - ref).abs().max().item()
    assert err < 2e-4 * max(1.0, K ** 0.5), f"{layout} {M}x{N}x{K}: max err {err}"


def test_gemm_epilogues_and_split_k(lib):
    import gpu_ops
    g = torch.Generator().manual_seed(5)
    M, N, K = 300, 150, 77
    A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
    bias, mask = torch.randn(N, generator=g), (torch.rand(M, N, generator=g) > 0.3).float() * 1.25
    C0 = torch.randn(M, N, generator=g)
    for act, fn in ((0, lambda x: x), (1, torch.relu), (2, torch.tanh)):
        Cd, Ad, Bd, bias_d, mask_d = dev(C0.clone()), dev(A), dev(B), dev(bias), dev(mask)
        gpu_ops.gemm((Ad, 0, K, 1), (Bd, 0, N, 1), (Cd, 0, N), M, N, K, alpha=0.5, beta=2.0, bias=bias_d,
                     act=act, mask=mask_d)
        ref = fn(0.5 * (A @ B) + 2.0 * C0 + bias) * mask
        assert (Cd.cpu() - ref).abs().max().item() < 1e-4, f"act {act}"
    # split-K accumulates into C with atomics (weight-gradient form: K is the long dimension)
    M, N, K = 100, 130, 4000
    A, B = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    Cd, Ad, Bd = dev(C0.clone()), dev(A), dev(B)
    gpu_ops.gemm((Ad, 0, 1, M), (Bd, 0, N, 1), (Cd, 0, N), M, N, K, beta=1.0, split_k=16)
    ref = C0.double() + A.t().double() @ B.double()
    assert (Cd.cpu().double() - ref).abs().max().item() < 2e-3
    # column-sliced operands (the step addresses slices of wider buffers in place)
    M, N, K, ld = 50, 40, 30, 100
    Abig, Bbig = torch.randn(M, ld, generator=g), torch.randn(N, ld, generator=g)
    Cbig = torch.zeros(M, ld)
    Cd, Ad, Bd = dev(Cbig), dev(Abig), dev(Bbig)
    gpu_ops.gemm((Ad, 10, ld, 1), (Bd, 20, 1, ld), (Cd, 5, ld), M, N, K)
    ref = Abig[:, 10:10 + K] @ Bbig[:, 20:20 + K].t()
    got = Cd.cpu()
    assert (got[:, 5:5 + N] - ref).abs().max().item() < 1e-4
    assert got[:, :5].abs().max().item() == 0 and got[:, 5 + N:].abs().max().item() == 0


@pytest.mark.parametrize("M,N,K,layout,split,sums", [
    (400, 300, 5120, "tn", 8, False),      # dW_ih: 4 x 3 macro tiles, the last ones partly dead
    (400, 100, 5120, "tn", 8, True),       # dW_hh + bias sums: the ones column is fragment 7's fifth column
    (100, 100, 2560, "tn", 8, True),
    (96, 128, 1000, "tn", 4, True),        # N a multiple of 16: the ones column opens a fragment of its own; K tail
    (100, 150, 9216, "tn", 8, False),      # dW_key_vis: 8-byte loads of the 150-wide feature rows
    (9, 100, 5120, "tn", 8, False),        # dW_h2o: one fragment of rows
    (5120, 200, 500, "nn", 1, False),      # dS +=: k-contiguous A, row-contiguous B
    (9216, 400, 150, "nt", 1, False),      # U_vis: both k-contiguous, 8-byte loads
    (2560, 25, 800, "nn", 8, False),       # dxe: two fragments of columns
])
def test_gemm_macro_tiles_with_slabs_are_exact_and_reproducible(lib, M, N, K, layout, split, sums):
    """gemm_mt.hip through gscan_gemm_f32_scratch: split-K partial tiles go to slabs that a second launch adds in a
    fixed order — the result equals the reference and is BITWISE the same from run to run (the atomics are not)."""
    import gpu_ops
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    s0 = torch.randn(M, generator=g)
    Ad = dev(A) if layout[0] == "n" else dev(A.t().contiguous()).t()
    Bd = dev(B) if layout[1] == "n" else dev(B.t().contiguous()).t()
    scratch = torch.full((48 << 20,) if split > 1 else (16,), float("nan"), device="cuda")     # NaN: nothing stale is ever added
    outs = []
    for _ in range(3):
        Cd, sd = dev(C0.clone()), dev(s0.clone())
        gpu_ops.gemm_scratch((Ad, 0, Ad.stride(0), Ad.stride(1)), (Bd, 0, Bd.stride(0), Bd.stride(1)), (Cd, 0, N), M, N, K,
                             beta=1.0, split_k=split, asum=sd if sums else None, scratch=scratch if split > 1 else None)
        outs.append((Cd.cpu(), sd.cpu()))
    ref = C0.double() + A.double() @ B.double()
    err = (outs[0][0].double() - ref).abs().max().item()
    assert err < 3e-4 * max(1.0, K ** 0.5), f"{layout} {M}x{N}x{K}: max err {err}"
    if sums:
        rs = s0.double() + A.double().sum(1)
        assert (outs[0][1].double() - rs).abs().max().item() < 3e-4 * K ** 0.5
    for c, sv in outs[1:]:
        assert torch.equal(c, outs[0][0]) and torch.equal(sv, outs[0][1]), "slab reduction is not reproducible"


@pytest.mark.parametrize("split", [1, 4])
@pytest.mark.parametrize("macro", [False, True])
def test_gemm_row_sums_are_not_scaled_by_alpha(lib, macro, split):
    """include/gscan_hip.h: asum[m] += sum_k A(m,k), whatever alpha multiplies the product with — on the 32 x 64 tiles
    (no scratch) and on the macro tiles (scratch handed in: atomics are replaced by slabs when split), both through
    gscan_gemm_f32_scratch.  (the macro tiles' ones column used to be scaled by alpha.)"""
    import gpu_ops
    M, N, K, alpha = 96, 80, 1024, -0.375
    g = torch.Generator().manual_seed(5)
    A, B, C0, s0 = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g), torch.randn(M, N, generator=g), torch.randn(M, generator=g)
    Ad, Bd, Cd, sd = dev(A).t(), dev(B), dev(C0.clone()), dev(s0.clone())
    scratch = torch.full((8 << 20,), float("nan"), device="cuda") if macro else None
    gpu_ops.gemm_scratch((Ad, 0, Ad.stride(0), Ad.stride(1)), (Bd, 0, Bd.stride(0), Bd.stride(1)), (Cd, 0, N), M, N, K,
                         alpha=alpha, beta=1.0, split_k=split, asum=sd, scratch=scratch)
    ref = C0.double() + alpha * (A.t().double() @ B.double())
    assert (Cd.cpu().double() - ref).abs().max().item() < 1e-3
    assert (sd.cpu().double() - (s0.double() + A.double().sum(0))).abs().max().item() < 1e-3


@pytest.mark.parametrize("M,N,K,layout,split", [
    (32768, 400, 600, "nt", 1),     # 256 x 4 macro tiles, K >= 512: the rule's launch; seven fragments of columns
    (32768, 400, 150, "nt", 1),     # as many tiles but a short K: stays on the small tiles (the training step's forward products)
    (32768, 100, 100, "nt", 1),     # too few tiles for the rule: rides on the small tiles
    (65536, 240, 518, "nn", 1),     # 512 x 2 tiles, row-contiguous B, 8-byte loads (rows of 518 floats)
    (4096, 4096, 515, "tn", 1),     # both operands row-contiguous (weight-gradient layout), ragged K
    (2048, 1040, 4096, "tn", 8),    # the rule's launch with split-K and no scratch: atomics on macro tiles
])
def test_gemm_large_launches_take_macro_tiles(lib, M, N, K, layout, split):
    """Launches with >= 1024 macro tiles of products at least 512 deep go to gemm_mt_kernel (csrc/gemm_mt.hip) by rule;
    same contract on both sides of the rule."""
    import gpu_ops
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    Ad = dev(A) if layout[0] == "n" else dev(A.t().contiguous()).t()
    Bd = dev(B) if layout[1] == "n" else dev(B.t().contiguous()).t()
    Cd = torch.zeros(M, N, device="cuda")
    kw = dict(beta=1.0, split_k=split) if split > 1 else {}
    gpu_ops.gemm((Ad, 0, Ad.stride(0), Ad.stride(1)), (Bd, 0, Bd.stride(0), Bd.stride(1)), (Cd, 0, N), M, N, K, **kw)
    ref = (dev(A).double() @ dev(B).double())
    err = (Cd.double() - ref).abs().max().item()
    assert err < 2e-4 * max(1.0, K ** 0.5), f"{layout} {M}x{N}x{K}: max err {err}"


@pytest.mark.parametrize("bm", ["64", "128"])
def test_gemm_suite_on_forced_macro_tiles(bm):
    """The layout / epilogue / split-K tests again in a child process in DETERMINISTIC mode (GSCAN_DETERMINISTIC=1:
    every product on the macro tiles whatever its size or operand layout — bias, activation, mask, beta, atomics
    without scratch, ragged edges in M, N and K, 4-byte loads), once per tile height."""
    import os, subprocess, sys
    env = dict(os.environ, GSCAN_DETERMINISTIC="1", GSCAN_MT_BM=bm)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "test_gemm_layouts or test_gemm_epilogues_and_split_k or test_gemm_short_k", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("M,N,K,lda,ldb,epilogue", [
    (9216, 400, 150, 150, 150, False),   # the visual gate images: contiguous rows of 150 floats (8-byte loads), ragged N tiles
    (5120, 400, 100, 400, 300, True),    # embedding part of the gates: column slices of wider buffers, bias
    (8200, 100, 152, 152, 160, True),    # ragged M, tanh + mask
    (6000, 70, 37, 40, 37, False),       # odd K: 4-byte loads
    (40000, 64, 96, 96, 96, False),      # K a whole number of 32-deep chunks: no tail steps
])
def test_gemm_short_k_products(lib, M, N, K, lda, ldb, epilogue):
    """Tall k-contiguous products with K <= 152 (the forward launch's shapes: column slices of wider buffers, ragged
    edges, epilogues) on whatever kernel the default rule picks; test_gemm_suite_on_forced_macro_tiles runs them on
    the macro tiles.  Same contract either way."""
    import gpu_ops
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    Abig, Bbig = torch.randn(M, lda, generator=g), torch.randn(N, ldb, generator=g)
    A, B = Abig[:, :K], Bbig[:, :K]
    bias, mask = torch.randn(N, generator=g), (torch.rand(M, N, generator=g) > 0.3).float() * 1.25
    Cd = torch.zeros(M, N, device="cuda")
    kw = dict(bias=dev(bias), act=2, mask=dev(mask)) if epilogue else {}
    gpu_ops.gemm((dev(Abig), 0, lda, 1), (dev(Bbig), 0, 1, ldb), (Cd, 0, N), M, N, K, **kw)
    ref = A.double() @ B.double().t()
    if epilogue:
        ref = torch.tanh(ref + bias.double()) * mask.double()
    err = (Cd.cpu().double() - ref).abs().max().item()
    assert err < 2e-4 * max(1.0, K ** 0.5), f"{M}x{N}x{K}: max err {err}"


WS_CASES = [  # M, N, K, lda, ldb, bias, act
    (9216, 400, 150, 150, 150, 0, 0),   # visual gate images: contiguous rows of 150 floats (8-byte fragment reads), five 80-column blocks
    (9216, 100, 150, 150, 150, 0, 0),   # visual keys: one block of seven fragments, 12 dead columns
    (5120, 400, 100, 400, 300, 1, 0),   # embedded gates: A and B are column slices of wider buffers, bias
    (2560, 100, 100, 100, 100, 0, 0),   # textual keys
    (256, 100, 100, 100, 100, 1, 2),    # bridge: four tiles, tanh
    (8200, 300, 152, 152, 160, 1, 2),   # ragged M (8200 = 128 x 64 + 8), 80 + 80 + 80 + 60 columns
    (1000, 9, 26, 26, 26, 0, 0),        # one fragment, a K of three 8-blocks and a bit, M not a multiple of 64
    (6000, 70, 36, 40, 40, 0, 0),       # strided A rows (16-byte aligned), K = 36
    (3, 17, 160, 160, 160, 1, 0),       # fewer rows than a tile, the deepest K
]


def test_gemm_weights_stationary_kernel():
    """gemm_ws.hip (GSCAN_GEMM_WS=3: every launch MUST take it, an ineligible one is an error) against float64 on the
    forward launch's shapes and on the edges of what it takes: ragged M and N, one fragment, short tiles, both
    fragment-read widths, and the step's own forward launch as ONE grouped launch (seven products, 19 column blocks)."""
    import os, subprocess, sys
    worker = r"""
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import gpu_ops
from test_kernels_gpu import WS_CASES
for M, N, K, lda, ldb, bias, act in WS_CASES:
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K)
    Abig, Bbig = torch.randn(M, lda, generator=g), torch.randn(N, ldb, generator=g)
    bv = torch.randn(N, generator=g)
    Cd = torch.full((M, N + 3), 7.0, device="cuda")                  # wider than N: nothing may be written past column N
    gpu_ops.gemm((Abig.cuda(), 0, lda, 1), (Bbig.cuda(), 0, 1, ldb), (Cd, 0, N + 3), M, N, K, bias=bv.cuda() if bias else None, act=act)
    torch.cuda.synchronize()
    ref = Abig[:, :K].double() @ Bbig[:, :K].double().t()
    if bias: ref = ref + bv.double()
    if act == 2: ref = torch.tanh(ref)
    out = Cd.cpu().double()
    err = (out[:, :N] - ref).abs().max().item()
    assert err < 2e-4 * max(1.0, K ** 0.5), (M, N, K, err)
    assert (out[:, N:] == 7.0).all(), (M, N, K)
    print("ok", M, N, K, err, flush=True)
print("all ok")
"""
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-c", worker, os.path.dirname(here), here],
                       env=dict(os.environ, GSCAN_GEMM_WS="3"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("B,G,Cc,K3,Co,density,u8", [
    (5, 6, 16, 7, 50, 0.2, False),        # the paper's shape, float32 world
    (37, 6, 16, 13, 50, 0.07, True),      # k = 13 (every cell reaches every cell), uint8 world, two backward slices
    (3, 4, 15, 7, 50, 1.0, False),        # a DENSE world: the sparse kernels are exact for any input
    (2, 15, 16, 7, 70, 0.05, True),       # the reference's 15 x 15 test grid, Co > 64 (two output-channel chunks)
    (4, 3, 5, 3, 7, 0.5, False),          # tiny everything
    (3, 6, 16, 7, 100, 0.15, True),       # 3 Co = 300 feature columns: more than one column group of the bias sums
])
def test_world_encoder_forward_and_weight_gradients(lib, B, G, Cc, K3, Co, density, u8):
    """The input-sparse world encoder equals the reference's three conv2d on the transposed image + ReLU + dropout
    mask (cnn_model.py:28-35), and its backward kernel equals autograd's kernel / bias gradients; float32 and uint8
    worlds, sparse and dense, non-{0,1} values included."""
    import ctypes as C
    import gpu_ops
    from multimodal_seq2seq_gscan_amd import _lib
    g = torch.Generator().manual_seed(3)
    M, F = G * G, 3 * Co
    world = (torch.rand(B, G, G, Cc, generator=g) < density).float()
    if u8:
        world = world * torch.randint(1, 4, world.shape, generator=g).float()      # bytes other than 1 widen exactly
    else:
        world = world * (0.5 + torch.rand(world.shape, generator=g))               # arbitrary float values
    world[0] = 0 if B > 1 else world[0]                                             # an example without non-zeros
    Ws = [(torch.randn(Co, Cc, k, k, generator=g) * 0.1).requires_grad_(True) for k in (1, 5, K3)]
    bs = [(torch.randn(Co, generator=g) * 0.1).requires_grad_(True) for _ in range(3)]
    mask = (torch.rand(B, M, F, generator=g) > 0.1).float() / 0.9
    ref = torch.cat([torch.nn.functional.conv2d(world.transpose(1, 3), W, b, padding=W.shape[-1] // 2).transpose(1, 3)
                     for W, b in zip(Ws, bs)], dim=3).reshape(B, M, F)
    feat_ref = torch.relu(ref) * mask
    d_out = torch.randn(B, M, F, generator=g)
    feat_ref.backward(d_out)
    world_d, mask_d = dev(world.to(torch.uint8) if u8 else world), dev(mask)
    W_d, b_d = [dev(W.detach()) for W in Ws], [dev(b.detach()) for b in bs]
    ptrs = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])
    scratch = torch.empty((26 + K3 * K3) * Cc * ((Co + 31) // 32 * 32), device="cuda")   # rows padded to 128 bytes
    feat = torch.empty(B, M, F, device="cuda")
    _lib.check(lib.gscan_world_encoder_forward(world_d.data_ptr(), int(u8), ptrs(W_d), ptrs(b_d), B, G, Cc, Co, K3,
                                               mask_d.data_ptr(), scratch.data_ptr(), feat.data_ptr(),
                                               gpu_ops.stream()), "world_encoder_forward")
    assert (feat.cpu() - feat_ref.detach()).abs().max().item() < 1e-4
    # d(pre-activation) = d_out * mask where ReLU was active (what the step's key-layer kernel hands over)
    dpre = dev(d_out * mask * (ref > 0).float())
    gW = [torch.zeros_like(W) for W in W_d]
    gb = [torch.zeros_like(b) for b in b_d]
    lists = torch.empty(lib.gscan_world_encoder_backward_scratch_floats(B, G, Cc), device="cuda")
    _lib.check(lib.gscan_world_encoder_backward(world_d.data_ptr(), int(u8), dpre.data_ptr(), B, G, Cc, Co, K3,
                                                lists.data_ptr(), ptrs(gW), ptrs(gb), gpu_ops.stream()),
               "world_encoder_backward")
    for W, b, gw_, gb_ in zip(Ws, bs, gW, gb):
        assert torch.allclose(gw_.cpu(), W.grad, atol=1e-4, rtol=1e-4), W.shape
        assert torch.allclose(gb_.cpu(), b.grad, atol=1e-4, rtol=1e-4)


def _lstm_reference(x, lengths, lstm):
    """nn.LSTM over packed rows, directions summed: the reference's encoder core (seq2seq_model.py:62-88)."""
    from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence
    B, L, _ = x.shape
    He = lstm.hidden_size
    packed = pack_padded_sequence(x, lengths.cpu(), batch_first=True, enforce_sorted=False)
    out, (h, _) = lstm(packed)
    out, _ = pad_packed_sequence(out, batch_first=True, total_length=L)
    D = 2 if lstm.bidirectional else 1
    return out.view(B, L, D, He).sum(2), h.view(1, D, B, He).sum(1)[0]


@pytest.mark.parametrize("He,bidir", [(20, True), (100, True), (100, False), (32, True)])
def test_encoder_lstm_forward_backward(lib, He, bidir):
    import gpu_ops
    from multimodal_seq2seq_gscan_amd import _lib
    torch.manual_seed(11)
    B, L, E = 9, 7, 12
    D = 2 if bidir else 1
    lstm = torch.nn.LSTM(E, He, bidirectional=bidir, batch_first=True)
    x = torch.randn(B, L, E, requires_grad=True)
    lengths = torch.tensor([7, 3, 5, 1, 7, 2, 6, 4, 7])
    out_ref, h_ref = _lstm_reference(x, lengths, lstm)
    d_out, d_h = torch.randn(B, L, He), torch.randn(B, He)
    live = (torch.arange(L)[None, :] < lengths[:, None]).float()[:, :, None]
    ((out_ref * d_out * live).sum() + (h_ref * d_h).sum()).backward()

    names = ["", "_reverse"][:D]
    w_ih = [getattr(lstm, "weight_ih_l0" + s).detach() for s in names]
    w_hh = [getattr(lstm, "weight_hh_l0" + s).detach() for s in names]
    b_ih = [getattr(lstm, "bias_ih_l0" + s).detach() for s in names]
    b_hh = [getattr(lstm, "bias_hh_l0" + s).detach() for s in names]
    gx = torch.stack([x.detach() @ w_ih[d].t() + b_ih[d] for d in range(D)], dim=2).contiguous()   # [B,L,D,4He]
    c = {k: torch.full(s, float("nan"), device="cuda") for k, s in dict(
        out=(B, L, He), hf=(B, He), gates=(B, L, D, 4 * He), cells=(B, L, D, He), hprev=(B, L, D, He),
        delta=(B, L, D, 4 * He)).items()}
    len_d = dev(lengths.int())
    wd = [dev(w) for w in w_hh] + [None]
    bd = [dev(b) for b in b_hh] + [None]
    gx_d, d_out_d, d_h_d = dev(gx), dev(d_out * live), dev(d_h)
    scratch = torch.empty(D * 4 * He * He, device="cuda")         # register image of W_hh, written by the call
    _lib.check(lib.gscan_encoder_lstm_forward(B, L, He, D, gx_d.data_ptr(), len_d.data_ptr(), wd[0].data_ptr(),
                                              bd[0].data_ptr(), _lib.ptr(wd[1]), _lib.ptr(bd[1]),
                                              c["out"].data_ptr(), c["hf"].data_ptr(), c["gates"].data_ptr(),
                                              c["cells"].data_ptr(), c["hprev"].data_ptr(), scratch.data_ptr(),
                                              gpu_ops.stream()), "fwd")
    assert (c["out"].cpu() - out_ref.detach()).abs().max().item() < 2e-5
    assert (c["hf"].cpu() - h_ref.detach()).abs().max().item() < 2e-5
    _lib.check(lib.gscan_encoder_lstm_backward(B, L, He, D, len_d.data_ptr(), wd[0].data_ptr(), _lib.ptr(wd[1]),
                                               c["gates"].data_ptr(), c["cells"].data_ptr(),
                                               d_out_d.data_ptr(), d_h_d.data_ptr(),
                                               c["delta"].data_ptr(), gpu_ops.stream()), "bwd")
    delta = c["delta"].cpu()
    assert torch.isfinite(delta).all()
    hprev = c["hprev"].cpu()
    assert torch.isfinite(hprev).all()
    for d, s in enumerate(names):
        dl = delta[:, :, d].reshape(B * L, 4 * He)
        g_whh = dl.t() @ hprev[:, :, d].reshape(B * L, He)
        g_wih = dl.t() @ x.detach().reshape(B * L, E)
        assert (g_whh - getattr(lstm, "weight_hh_l0" + s).grad).abs().max().item() < 1e-4, "w_hh" + s
        assert (g_wih - getattr(lstm, "weight_ih_l0" + s).grad).abs().max().item() < 1e-4, "w_ih" + s
        assert (dl.sum(0) - getattr(lstm, "bias_hh_l0" + s).grad).abs().max().item() < 1e-4, "bias" + s
    dx = sum(delta[:, :, d] @ w_ih[d] for d in range(D))
    assert (dx - x.grad).abs().max().item() < 1e-4


def test_losses_metrics_adam_dropout(lib):
    import gpu_ops
    from multimodal_seq2seq_gscan_amd import _lib
    from oracle import seq2seq_oracle as oracle
    g = torch.Generator().manual_seed(2)
    B, T, V = 7, 9, 6
    logp = torch.log_softmax(torch.randn(B, T, V, generator=g), -1)
    targets = torch.randint(3, V, (B, T), generator=g)
    targets[:, 0] = 1
    for b in range(B):
        n = int(torch.randint(3, T + 1, (1,), generator=g))
        targets[b, n - 1] = 2
        targets[b, n:] = 0
    out = torch.zeros(2, device="cuda")
    dl = torch.empty(B, T, V, device="cuda")
    logp_d, targets_d = dev(logp), dev(targets)
    _lib.check(lib.gscan_sequence_nll(logp_d.data_ptr(), targets_d.data_ptr(), B, T, V, 0, out.data_ptr(),
                                      out.data_ptr() + 4, dl.data_ptr(), gpu_ops.stream()), "nll")
    total, n = oracle.sequence_loss(logp, targets, 0, reduction="sum")
    assert abs(out[0].item() - total.item()) < 1e-4 and out[1].item() == n.item()
    lp = logp.clone().requires_grad_(True)
    oracle.sequence_loss(lp, targets, 0, reduction="sum")[0].backward()
    assert torch.equal(dl.cpu(), lp.grad)
    m = torch.zeros(3, device="cuda")
    _lib.check(lib.gscan_sequence_metrics(logp_d.data_ptr(), targets_d.data_ptr(), B, T, V, 0, m.data_ptr(),
                                          gpu_ops.stream()), "metrics")
    acc, exact = oracle.metrics(logp, targets)
    c, live, ex = m.tolist()
    assert abs(100 * c / live - acc) < 1e-4 and abs(100 * ex / B - exact) < 1e-4
    # auxiliary NLL
    aux = torch.log_softmax(torch.randn(B, 16, generator=g), -1)
    pos = torch.randint(0, 16, (B,), generator=g)
    o1 = torch.zeros(1, device="cuda")
    da = torch.empty(B, 16, device="cuda")
    aux_d, pos_d = dev(aux), dev(pos)
    _lib.check(lib.gscan_position_nll(aux_d.data_ptr(), pos_d.data_ptr(), B, 16, o1.data_ptr(), da.data_ptr(),
                                      gpu_ops.stream()), "pos nll")
    assert abs(o1.item() / B - oracle.auxiliary_loss(aux, pos).item()) < 1e-5
    assert da.cpu().sum().item() == -B
    # fused Adam == the oracle's restatement of torch.optim.Adam + LambdaLR
    n = 10007
    p0, g0 = torch.randn(n, generator=g), torch.randn(n, generator=g) * 0.01
    p_ref, m_ref, v_ref = [p0.clone()], [torch.zeros(n)], [torch.zeros(n)]
    pd, md, vd = dev(p0.clone()), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in (1, 2, 3):
        gs = g0 * step
        oracle.adam_step(p_ref, [gs], m_ref, v_ref, step, 1e-3, lr_decay=0.9, lr_decay_steps=2.0)
        gs_d = dev(gs)
        _lib.check(lib.gscan_adam_step(pd.data_ptr(), gs_d.data_ptr(), md.data_ptr(), vd.data_ptr(), n, 1e-3, 0.9,
                                       0.999, 1e-8, 0.9, 2.0, step, None, gpu_ops.stream()), "adam")
    assert (pd.cpu() - p_ref[0]).abs().max().item() < 2e-6
    # grad_scale read from device memory
    pd2, md2, vd2 = dev(p0.clone()), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    scale = torch.tensor([0.25], device="cuda")
    g4_d = dev(g0 * 4)
    _lib.check(lib.gscan_adam_step(pd2.data_ptr(), g4_d.data_ptr(), md2.data_ptr(), vd2.data_ptr(), n, 1e-3,
                                   0.9, 0.999, 1e-8, 0.9, 2.0, 1, scale.data_ptr(), gpu_ops.stream()), "adam")
    p_one = [p0.clone()]
    oracle.adam_step(p_one, [g0], [torch.zeros(n)], [torch.zeros(n)], 1, 1e-3, lr_decay=0.9, lr_decay_steps=2.0)
    assert (pd2.cpu() - p_one[0]).abs().max().item() < 2e-6
    # Philox dropout masks: values in {0, 1/(1-p)}, keep rate ~ 1-p, reproducible, streams differ
    nmask, p = 1 << 20, 0.3
    a, b2, c2 = (torch.empty(nmask, device="cuda") for _ in range(3))
    for t, sid in ((a, 0), (b2, 0), (c2, 1)):
        _lib.check(lib.gscan_dropout_mask(t.data_ptr(), nmask, p, 42, sid, gpu_ops.stream()), "mask")
    assert torch.equal(a, b2) and not torch.equal(a, c2)
    vals = torch.unique(a.cpu())
    assert len(vals) == 2 and vals[0] == 0 and abs(vals[1].item() - 1 / 0.7) < 1e-6
    assert abs((a != 0).float().mean().item() - 0.7) < 3e-3
