import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from multimodal_seq2seq_gscan_amd import _lib
from multimodal_seq2seq_gscan_amd.config import model_kwargs
from multimodal_seq2seq_gscan_amd.model import Model
from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lib = _lib.load()
cfg = model_kwargs("compositional", encoder_hidden_size=min(H, 128), decoder_hidden_size=H)
model = Model(**cfg).cuda().eval()
batch = {k: v.cuda() for k, v in make_batch(Shape(batch=256), 1).items()}
lib.gscan_probe_enable(2)
for _ in range(3):
    model.zero_grad()
    logp, _ = model(commands_input=batch["commands"], commands_lengths=batch["cmd_lengths"].tolist(), situations_input=batch["world"],
                    target_batch=batch["targets"], target_lengths=batch["tgt_lengths"].tolist())
torch.cuda.synchronize()
B, L = batch["commands"].shape
dims = model._dims(B, L, 20, batch["world"].shape[1])
st = model.workspace_view(dims, "stamps").cpu()[:16].tolist()
names = ["rows h (W_qt W_hh W_q2k_h)", "scores text", "softmax text", "cols text (ctx, U_t, U2_t | W_q2k)", "rows W_qv", "scores vis", "softmax vis", "cols ctx_vis", "cols U_v | rows W_ih ctx", "gates"]
tot = sum(st[:10])
print(f"H={H} forward streaming kernel, workgroup 0: {tot / 20:.0f} cycles per step")
for n, v in zip(names, st): print(f"  {n:38s} {v / 20:8.0f}")
