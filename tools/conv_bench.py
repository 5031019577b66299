"""Diagnostic: the input-sparse world encoder kernels alone (gscan_world_encoder_forward / _backward through the C
ABI) on benchmark-shaped synthetic worlds; HIP-event time per call, float32 and uint8 worlds."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import torch

from multimodal_seq2seq_gscan_amd import _lib
from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch

lib = _lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
ptrs = lambda ts: (C.c_void_p * 3)(*[t.data_ptr() for t in ts])


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for (B, G, K3, dense) in [(256, 6, 7, False), (64, 6, 7, False)] if os.environ.get('GSCAN_CONV_DEBUG') else [(256, 6, 7, False), (256, 6, 13, False), (64, 6, 7, False), (256, 6, 7, True)]:
    Cc, Co = 16, 50
    M, F = G * G, 3 * Co
    world = make_batch(Shape(batch=B, grid=G, channels=Cc), 5)["world"]
    if dense:
        world = torch.ones_like(world)
    nnz = (world != 0).sum().item() / B
    Ws = [torch.randn(Co, Cc, k, k, device="cuda") * 0.1 for k in (1, 5, K3)]
    bs = [torch.randn(Co, device="cuda") * 0.1 for _ in range(3)]
    img = torch.empty((26 + K3 * K3) * Cc * ((Co + 31) // 32 * 32), device="cuda")
    feat = torch.empty(B, M, F, device="cuda")
    dfeat = torch.randn(B, M, F, device="cuda")
    gW, gb = [torch.zeros_like(W) for W in Ws], [torch.zeros_like(b) for b in bs]
    lists = torch.empty(lib.gscan_world_encoder_backward_scratch_floats(B, G, Cc), device="cuda")
    for u8 in (False, True):
        wd = (world.to(torch.uint8) if u8 else world).cuda()
        fwd = lambda: _lib.check(lib.gscan_world_encoder_forward(wd.data_ptr(), int(u8), ptrs(Ws), ptrs(bs), B, G, Cc,
                                                                Co, K3, None, img.data_ptr(), feat.data_ptr(), st()), "f")
        bwd = lambda: _lib.check(lib.gscan_world_encoder_backward(wd.data_ptr(), int(u8), dfeat.data_ptr(), B, G, Cc, Co,
                                                                 K3, lists.data_ptr(), ptrs(gW), ptrs(gb), st()), "b")
        print(f"B={B} G={G} k={K3} nnz/example={nnz:.1f} {'u8 ' if u8 else 'f32'}: forward (image + conv) "
              f"{timed(fwd):7.1f} us, backward {timed(bwd):7.1f} us", flush=True)
