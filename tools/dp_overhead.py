"""Diagnostic: what the data-parallel form of the step costs on ONE device — the step's collectives issued for real on
a one-rank RCCL communicator (backend "nccl") against the single-process fused step.  The transfer itself is absent
(one rank); what shows is the launch / stream hand-over cost of RCCL and of the sum-loss backward + mean Adam form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from multimodal_seq2seq_gscan_amd.config import model_kwargs
from multimodal_seq2seq_gscan_amd.model import Model
from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
from multimodal_seq2seq_gscan_amd.train import TrainStep

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
for aux in (False, True):
    cfg = model_kwargs("compositional", auxiliary_task=aux)
    batch = {k: v.cuda() for k, v in make_batch(Shape(batch=256), 1).items()}
    for collective, native, buckets in ((False, None, 1), (True, False, 1), (True, True, 1), (True, True, 2)):
        torch.manual_seed(0)
        step = TrainStep(Model(**cfg).cuda(), always_collective=collective, native_allreduce=native, dp_buckets=buckets)
        for _ in range(20):
            step(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            step(batch)
        torch.cuda.synchronize()
        how = "none" if not collective else ("gscan_allreduce_f32 on the step's stream (1 rank)" if step.exchange.comm is not None
                                              else "torch.distributed RCCL stream (1 rank)")
        if collective and buckets == 2:
            how += ", TWO buckets (early group all-reduced by the backward pass on its first leaf stream: gscan_comm_set_early_allreduce)"
        print(f"auxiliary={aux} collectives={how}: "
              f"{1e3 * (time.perf_counter() - t0) / 100:.4f} ms/step", flush=True)
dist.destroy_process_group()
