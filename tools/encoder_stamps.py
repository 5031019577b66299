"""Experiment: where the command encoder's forward workgroup (0,0) spends its cycles (needs -DGSCAN_ENC_STAMPS):
    python tools/variants.py est:all:-DGSCAN_ENC_STAMPS,-DGSCAN_TRACE && GSCAN_HIP_LIB=variants/libgscan_hip.est.so python tools/encoder_stamps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from multimodal_seq2seq_gscan_amd import _lib
from multimodal_seq2seq_gscan_amd.config import model_kwargs
from multimodal_seq2seq_gscan_amd.model import Model
from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
from multimodal_seq2seq_gscan_amd.train import TrainStep
lib = _lib.load()
model = Model(**model_kwargs("compositional")).cuda()
batch = {k: v.cuda() for k, v in make_batch(Shape(batch=256), 1).items()}
batch["cmd_lengths"] = batch["cmd_lengths"].to(torch.int32)
step = TrainStep(model)
for _ in range(5):
    step(batch)
torch.cuda.synchronize()
buf = torch.zeros(2 + 6 * 256, dtype=torch.int64, device="cuda")
_lib.check(lib.gscan_trace_set(buf.data_ptr()), "trace_set")
reps = 10
for _ in range(reps):
    step(batch)
torch.cuda.synchronize()
_lib.check(lib.gscan_trace_set(None), "trace_set")
t = buf.cpu().tolist()[1520:1525]
names = ["stage x", "input projection", "pad + recurrent weights", "recurrence", "write-back"]
print("encoder forward, workgroup (0,0): " + "  ".join(f"{n}={v / reps:.0f}" for n, v in zip(names, t)) + f"  total={sum(t) / reps:.0f} cycles")
