"""Diagnostic: where the batcher-fed training loop spends host time (stage / deliver / step issue) and what the GPU
does meanwhile."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_seq2seq_gscan_amd.config import model_kwargs
from multimodal_seq2seq_gscan_amd.dataset import BatchStager, GroundedScanDataset
from multimodal_seq2seq_gscan_amd.model import Model
from multimodal_seq2seq_gscan_amd.synthetic import Shape, write_dataset_file
from multimodal_seq2seq_gscan_amd.train import TrainStep

tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "dataset.txt")
write_dataset_file(path, {"train": 20000}, Shape(batch=1), seed=7)
data = GroundedScanDataset(path, tmp, k=0, split="train", generate_vocabulary=True)
data.read_dataset()
cfg = model_kwargs("compositional", input_vocabulary_size=data.input_vocabulary_size,
                   target_vocabulary_size=data.target_vocabulary_size)
model = Model(**cfg).cuda()
step = TrainStep(model)
B = 256
stager = BatchStager(torch.device("cuda"), data.slab_bytes(B))
keys = ("commands", "cmd_lengths", "world", "targets", "tgt_lengths", "target_positions")
for mode in ("stage only", "stage + step", "stage + step (sync each)"):
    data.shuffle_data()
    t_stage = t_step = 0.0
    n = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = data.batches(B, stager=stager)
    while True:
        a = time.perf_counter()
        b = next(it, None)
        c = time.perf_counter()
        if b is None or b["commands"].shape[0] != B:
            break
        if mode != "stage only":
            step({k: b[k] for k in keys})
            if mode.endswith("each)"):
                torch.cuda.synchronize()
        d = time.perf_counter()
        t_stage += c - a
        t_step += d - c
        n += 1
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{mode:28s}: {n} batches, wall {1e3 * el / n:.3f} ms/batch, next() {1e3 * t_stage / n:.3f}, step() {1e3 * t_step / n:.3f}")

# ---- finer: wrap the torch / numpy calls of the stager
import collections
import numpy as np
acc = collections.defaultdict(float)
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); acc[name] += time.perf_counter() - t; return r
    return w
for i in range(stager.depth):
    stager.copied[i].synchronize = timed("copied.synchronize", stager.copied[i].synchronize)
    stager.copied[i].record = timed("copied.record", stager.copied[i].record)
stager.copy_stream.wait_event = timed("copy.wait_event", stager.copy_stream.wait_event)
orig_take = np.take
np.take = timed("np.take", orig_take)
orig_stage, orig_deliver = stager.stage, stager.deliver
stager.stage = timed("stage total", orig_stage)
stager.deliver = timed("deliver total", orig_deliver)
data.shuffle_data()
n = 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in data.batches(B, stager=stager):
    if b["commands"].shape[0] != B:
        break
    t = time.perf_counter(); step({k: b[k] for k in keys}); acc["step"] += time.perf_counter() - t
    n += 1
torch.cuda.synchronize()
print(f"wall {1e3 * (time.perf_counter() - t0) / n:.3f} ms/batch;", {k: round(1e3 * v / n, 3) for k, v in acc.items()})

np.take = orig_take
data.shuffle_data()
it = data.batches(B, stager=stager)
samples = []
for _ in range(4):
    b = next(it)
    samples.append({k: b[k].clone() for k in keys})
for s in samples:
    for _ in range(10):
        step(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        step(s)
    torch.cuda.synchronize()
    print("resident", tuple(s["commands"].shape), tuple(s["targets"].shape), f"{1e3 * (time.perf_counter() - t0) / 50:.3f} ms/step")
# alternate between the four resident batches (shape changes every step, no staging)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(100):
    step(samples[i % 4])
torch.cuda.synchronize()
print("alternating resident batches", f"{1e3 * (time.perf_counter() - t0) / 100:.3f} ms/step")
