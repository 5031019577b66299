"""Timeline of ONE undisturbed training step from in-kernel clock stamps (no profiler: rocprofv3 adds 6-10 us per
kernel boundary and makes the host the limiter of the forward prelude).

    python tools/device_timeline.py [--steps-before 30] [--single-stream]

Every kernel's first workgroup stamps the 100 MHz device clock on entry and its last workgroup on exit
(gscan_trace_set, csrc/common.h); starts and ends are matched per (kernel id, grid size) in order.  The stamps
live in a -DGSCAN_TRACE build of the library: build it first (here or in the authoring container, it travels with
gpurun):  python tools/variants.py trace:all:-DGSCAN_TRACE
"""
import argparse
import os
import sys

ap = argparse.ArgumentParser()
ap.add_argument("--steps-before", type=int, default=30)
ap.add_argument("--single-stream", action="store_true")
ap.add_argument("--traced-steps", type=int, default=2, help="trace this many consecutive steps, print the last")
ap.add_argument("--workload", default="compositional", choices=["compositional", "target_length"],
                help="target_length = S3: k = 13, T = 120")
args = ap.parse_args()
if args.single_stream:
    os.environ["GSCAN_SINGLE_STREAM"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GSCAN_HIP_LIB", os.path.join(ROOT, "variants", "libgscan_hip.trace.so"))
if not os.path.exists(os.environ["GSCAN_HIP_LIB"]):
    raise SystemExit("build the traced library first: python tools/variants.py trace:all:-DGSCAN_TRACE")
import torch

from multimodal_seq2seq_gscan_amd import _lib
from multimodal_seq2seq_gscan_amd.config import model_kwargs
from multimodal_seq2seq_gscan_amd.model import Model
from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
from multimodal_seq2seq_gscan_amd.train import TrainStep

NAMES = {1: "dropout_mask", 2: "conv_fwd", 3: "prologue", 4: "gemm", 5: "encoder_fwd", 6: "decoder_fwd",
         7: "decoder_bwd", 8: "keys_backward", 9: "head_grad_finish", 10: "embed_grad", 11: "encoder_bwd",
         12: "conv_bwd", 13: "adam", 14: "loss", 15: "other"}
RECORDS = 256

lib = _lib.load()
torch.manual_seed(0)
model = Model(**model_kwargs(args.workload)).cuda()
step = TrainStep(model)
shape = Shape(batch=256, input_vocab=17, target_vocab=8, max_target=120) if args.workload == "target_length" else Shape(batch=256)
batch = {k: v.cuda() for k, v in make_batch(shape, 1).items()}
for _ in range(args.steps_before):
    step(batch)
torch.cuda.synchronize()
buf = torch.zeros(2 + 2 * 3 * RECORDS, dtype=torch.int64, device="cuda")
_lib.check(lib.gscan_trace_set(buf.data_ptr()), "gscan_trace_set")
for _ in range(args.traced_steps):
    step(batch)
torch.cuda.synchronize()
_lib.check(lib.gscan_trace_set(None), "gscan_trace_set")
t = buf.cpu().tolist()
ns, ne = min(t[0], RECORDS), min(t[1], RECORDS)
starts = [tuple(t[2 + 3 * i: 5 + 3 * i]) for i in range(ns)]
ends = [tuple(t[2 + 3 * RECORDS + 3 * i: 5 + 3 * RECORDS + 3 * i]) for i in range(ne)]
starts.sort(key=lambda r: r[2])
ends.sort(key=lambda r: r[2])
used = [False] * len(ends)
rows = []
for kid, grid, t0 in starts:
    t1 = None
    for j, (k2, g2, te) in enumerate(ends):
        if not used[j] and k2 == kid and g2 == grid and te >= t0:
            used[j] = True
            t1 = te
            break
    rows.append((t0, t1, kid, grid))
# the last traced step = everything behind the second-to-last optimiser launch (a step ends with adam, which since
# round 3 also draws the next step's dropout masks; a stand-alone dropout_mask launch opens a step only when the batch
# shape changed)
adams = [i for i, r in enumerate(rows) if r[2] == 13]
rows = rows[adams[-2] + 1:] if len(adams) >= 2 else rows
base = rows[0][0]
print(f"{'start us':>9} {'dur us':>8}  kernel (grid)")
for t0, t1, kid, grid in rows:
    dur = f"{(t1 - t0) / 100.0:8.1f}" if t1 is not None else "       ?"
    print(f"{(t0 - base) / 100.0:9.1f} {dur}  {NAMES.get(kid, kid)} ({grid})")
last_end = max(r[1] for r in rows if r[1] is not None)
print(f"span {(last_end - base) / 100.0:.1f} us, {len(rows)} kernels")
