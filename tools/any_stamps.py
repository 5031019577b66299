import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from multimodal_seq2seq_gscan_amd import _lib
from multimodal_seq2seq_gscan_amd.config import model_kwargs
from multimodal_seq2seq_gscan_amd.model import Model
from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
H = int(sys.argv[1]) if len(sys.argv) > 1 else 128
G = int(sys.argv[2]) if len(sys.argv) > 2 else 6          # python tools/any_stamps.py <hidden> [grid] [command length]
LC = int(sys.argv[3]) if len(sys.argv) > 3 else 10
lib = _lib.load()
cfg = model_kwargs("compositional", encoder_hidden_size=min(H, 128), decoder_hidden_size=H)
model = Model(**cfg).cuda().eval()
batch = {k: v.cuda() for k, v in make_batch(Shape(batch=256, grid=G, max_command=LC), 1).items()}
lib.gscan_probe_enable(2)
for _ in range(3):
    model.zero_grad()
    logp, _ = model(commands_input=batch["commands"], commands_lengths=batch["cmd_lengths"].tolist(), situations_input=batch["world"],
                    target_batch=batch["targets"], target_lengths=batch["tgt_lengths"].tolist())
    model.get_loss(logp, batch["targets"]).backward()
torch.cuda.synchronize()
B, L = batch["commands"].shape
dims = model._dims(B, L, 20, batch["world"].shape[1])
both = model.workspace_view(dims, "stamps").cpu().tolist()
st = both[:16]
names = ["rows h (W_qt W_hh W_q2k_h)", "scores text", "softmax text", "cols text (ctx, U_t, U2_t | W_q2k)", "rows W_qv", "scores vis", "softmax vis", "cols ctx_vis", "cols U_v | rows W_ih ctx", "gates"]
tot = sum(st[:10])
print(f"H={H} grid={G} L={LC} forward streaming kernel, workgroup 0: {tot / 20:.0f} cycles per step")
for n, v in zip(names, st): print(f"  {n:38s} {v / 20:8.0f}")
bw = both[16:32]
names_b = ["cell backward (saved activations from global)", "d ctx + alpha_vis, q_vis loads", "d alpha_vis rows (U_v, PK_v)", "visual attention backward",
           "W_qv^T (+ W_q2k ctx), alpha_text, q_text loads", "d alpha_text rows (U_t, U2_t, PK_t)", "textual attention backward", "W_hh^T",
           "W_qt^T, W_q2k_h^T / W_qv^T"]
print(f"H={H} backward streaming kernel, workgroup 0: {sum(bw[:9]) / 20:.0f} cycles per step")
for n, v in zip(names_b, bw): print(f"  {n:52s} {v / 20:8.0f}")
