"""Time of one training step (train.TrainStep: forward + loss + backward + Adam, one library call) over model shapes:
the benchmark shape on the register/LDS-resident kernels, and the shapes only the streaming kernels take
(DESIGN.md §4.1's table comes from this).  Usage: python tools/shape_sweep.py [--steps 30] [--batch 256]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from multimodal_seq2seq_gscan_amd.config import model_kwargs  # noqa: E402
from multimodal_seq2seq_gscan_amd.model import Model  # noqa: E402
from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch  # noqa: E402
from multimodal_seq2seq_gscan_amd.train import TrainStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--warmup", type=int, default=10)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--only", default="", help="comma-separated case names")
args = ap.parse_args()

# name, model overrides, shape overrides
CASES = [
    ("S1 hidden100 grid6 L10 T20", {}, {}),
    ("hidden128", dict(encoder_hidden_size=128, decoder_hidden_size=128), {}),
    ("hidden200", dict(encoder_hidden_size=200, decoder_hidden_size=200), {}),
    ("hidden256", dict(encoder_hidden_size=256, decoder_hidden_size=256), {}),
    ("hidden144", dict(encoder_hidden_size=144, decoder_hidden_size=144), {}),
    ("hidden160", dict(encoder_hidden_size=160, decoder_hidden_size=160), {}),
    ("hidden100 grid12", {}, dict(grid=12)),
    ("hidden100 command128", {}, dict(max_command=128)),
    ("hidden100 encoder256", dict(encoder_hidden_size=256), {}),
    ("hidden256 grid12 command128", dict(encoder_hidden_size=256, decoder_hidden_size=256), dict(grid=12, max_command=128)),
]
only = [s for s in args.only.split(",") if s]
for name, overrides, shape_kw in CASES:
    if only and not any(o in name for o in only):
        continue
    torch.manual_seed(0)
    cfg = model_kwargs("compositional", **overrides)
    model = Model(**cfg).cuda()
    kw = dict(batch=args.batch, input_vocab=cfg["input_vocabulary_size"], target_vocab=cfg["target_vocabulary_size"])
    kw.update(shape_kw)
    batch = make_batch(Shape(**kw), seed=3)
    dev = {k: v.cuda() for k, v in batch.items() if k in ("commands", "cmd_lengths", "world", "targets", "target_positions")}
    dev["world"] = dev["world"].to(torch.uint8)
    step = TrainStep(model, learning_rate=1e-3, weight_target_loss=0.3)
    for _ in range(args.warmup):
        out = step(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(dev)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    print(f"{name:36s} B={args.batch}: {ms:8.3f} ms/step  {args.batch / ms * 1e3:10.0f} examples/s  loss {out['loss'].item():.4f}",
          flush=True)
    step.close()
    del step, model
