"""Diagnostic: per-phase cycle shares of the two decoder kernels (in-kernel s_memtime stamps of workgroup 0).
    python tools/decoder_stamps.py [--target-length 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from multimodal_seq2seq_gscan_amd import _lib
from multimodal_seq2seq_gscan_amd.config import model_kwargs
from multimodal_seq2seq_gscan_amd.model import Model
from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch

ap = argparse.ArgumentParser()
ap.add_argument("--target-length", type=int, default=20)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--config", default="compositional", choices=["compositional", "demo"])
args = ap.parse_args()
lib = _lib.load()
cfg = model_kwargs(args.config)
model = Model(**cfg).cuda().eval()
shape = Shape(batch=args.batch, max_target=args.target_length)
if args.config == "demo":
    from multimodal_seq2seq_gscan_amd.synthetic import S0_DEMO
    import dataclasses
    shape = dataclasses.replace(S0_DEMO, batch=args.batch, max_target=args.target_length)
batch = {k: v.cuda() for k, v in make_batch(shape, 1).items()}
lib.gscan_probe_enable(2)
for _ in range(3):
    model.zero_grad()
    logp, _ = model(commands_input=batch["commands"], commands_lengths=batch["cmd_lengths"].tolist(),
                    situations_input=batch["world"], target_batch=batch["targets"],
                    target_lengths=batch["tgt_lengths"].tolist())
    model.get_loss(logp, batch["targets"]).backward()
torch.cuda.synchronize()
B, L = batch["commands"].shape
dims = model._dims(B, L, args.target_length, batch["world"].shape[1])
st = model.workspace_view(dims, "stamps").cpu()
for name, row in (("forward", st[:16]), ("backward", st[16:32])):
    tot = row.sum().item()   # slot 0 of the first step includes nothing else: once-per-launch pieces are in slots 10..
    print(name, "total cycles (s_memtime @100MHz ticks?)", tot, "per step", tot / args.target_length)
    print("   ", " ".join(f"{i}:{v / tot * 100:.1f}%" for i, v in enumerate(row.tolist()) if v > 0))
    print("   ", " ".join(f"{i}:{v / args.target_length:.0f}" for i, v in enumerate(row.tolist()[:10]) if v > 0))
    print("    once per launch (slots 10..):", " ".join(f"{i}:{v:.0f}" for i, v in enumerate(row.tolist()) if i >= 10 and v > 0))
