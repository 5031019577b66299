"""Diagnostic: the batcher-fed loop with length-bucketed batches — host time in next() / step(), wall per batch, and
the GPU time of each distinct batch shape (sync after every step)."""
import collections, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodal_seq2seq_gscan_amd.config import model_kwargs
from multimodal_seq2seq_gscan_amd.dataset import BatchStager, GroundedScanDataset
from multimodal_seq2seq_gscan_amd.model import Model
from multimodal_seq2seq_gscan_amd.synthetic import Shape, write_dataset_file
from multimodal_seq2seq_gscan_amd.train import TrainStep

tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "dataset.txt")
write_dataset_file(path, {"train": 60000}, Shape(batch=1), seed=7)
data = GroundedScanDataset(path, tmp, k=0, split="train", generate_vocabulary=True)
data.read_dataset()
cfg = model_kwargs("compositional", input_vocabulary_size=data.input_vocabulary_size,
                   target_vocabulary_size=data.target_vocabulary_size)
model = Model(**cfg).cuda()
step = TrainStep(model)
B = 256
stager = BatchStager(torch.device("cuda"), data.slab_bytes(B))
keys = ("commands", "cmd_lengths", "world", "targets", "tgt_lengths", "target_positions")
for bucket in (0, 8, 8):
    for sync in (False, True):
        data.shuffle_data(bucket_batches=bucket, batch_size=B)
        t_next = t_step = 0.0
        n = 0
        per_shape = collections.defaultdict(list)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it = data.batches(B, stager=stager)
        while True:
            a = time.perf_counter()
            b = next(it, None)
            c = time.perf_counter()
            if b is None or b["commands"].shape[0] != B:
                break
            step({k: b[k] for k in keys})
            if sync:
                torch.cuda.synchronize()
            d = time.perf_counter()
            t_next += c - a
            t_step += d - c
            if sync:
                per_shape[(b["commands"].shape[1], b["targets"].shape[1])].append(d - c)
            n += 1
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"bucket={bucket} sync={sync}: {n} batches, wall {1e3 * el / n:.3f} ms/batch, next() {1e3 * t_next / n:.3f}, step() {1e3 * t_step / n:.3f}")
        if sync and bucket:
            for shp in sorted(per_shape):
                v = sorted(per_shape[shp])
                print(f"    L={shp[0]:2d} T={shp[1]:2d}: {len(v):3d} steps, median {1e3 * v[len(v) // 2]:.3f} ms, max {1e3 * v[-1]:.3f}")
