"""Soak: N training steps (train.TrainStep, S1 shape, rotating batches) — loss finite throughout, the fused prologue launch's
self-service counter (conv.hip: chunks of the convolution weight image written by a waiting workgroup) reported at the end.
    python tools/soak.py [--steps 40000]"""
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
ap.add_argument("--steps", type=int, default=40000)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--target-length", type=int, default=0, help="0: the S1 shape; e.g. 120: long targets (S3; sums over time, DESIGN 4.5a)")
args = ap.parse_args()
torch.manual_seed(0)
cfg = model_kwargs("compositional", auxiliary_task=True)
model = Model(**cfg).cuda()
shape = Shape(batch=args.batch, input_vocab=cfg["input_vocabulary_size"], target_vocab=cfg["target_vocabulary_size"], ragged=True,
              **({"max_target": args.target_length} if args.target_length else {}))
batches = []
for k in range(8):
    b = {key: v.cuda() for key, v in make_batch(shape, seed=100 + k).items()
         if key in ("commands", "cmd_lengths", "world", "targets", "target_positions")}
    b["world"] = b["world"].to(torch.uint8)
    batches.append(b)
step = TrainStep(model, learning_rate=1e-3, weight_target_loss=0.3)
B, L = batches[0]["commands"].shape
dims = model._dims(B, L, batches[0]["targets"].shape[1], batches[0]["world"].shape[1])
t0 = time.perf_counter()
worst = 0.0
for i in range(args.steps):
    out = step(batches[i % len(batches)])
    if (i + 1) % (5000 if not args.target_length else 1000) == 0:
        loss = out["loss"].item()
        assert loss == loss and abs(loss) < 1e4, loss
        worst = max(worst, loss)
        print(f"step {i + 1}: loss {loss:.4f}  {1e3 * (time.perf_counter() - t0) / (i + 1):.4f} ms/step", flush=True)
torch.cuda.synchronize()
served = int(model.workspace_view(dims, "conv_flags").view(torch.int32)[512].item())
print(f"{args.steps} steps, self-served image chunks: {served}, parameters finite: {bool(torch.isfinite(model.flat_parameters).all())}")
step.close()
