"""Checkpoint-interop fixture, produced by the REFERENCE (authoring container only; the reference is imported,
never copied):  python tests/golden/make_golden_checkpoint.py  ->  tests/golden/demo_reference_checkpoint.pth.tar
                python tests/golden/make_golden_checkpoint.py --logp-only  ->  demo_reference_checkpoint_logp.npz
                (the REFERENCE loads the committed checkpoint with its own Model.load_model and scores one seeded
                batch: log-probabilities + loss, the known answer for `load_model` -> `Model.forward` on the device)

A checkpoint written by the reference's own Model.save_checkpoint after two Adam steps at the README demo
dimensions (model.py:246-261), plus — checked here, at generation time — the reverse direction: a checkpoint
written by this repository's Model loads into the reference Model and torch.optim.Adam."""
from __future__ import annotations

import os
import shutil
import sys
import tempfile
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")

from seq2seq.model import Model as ReferenceModel  # noqa: E402  (the reference, read-only)

from multimodal_seq2seq_gscan_amd.config import model_kwargs  # noqa: E402
from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch  # noqa: E402
from weights import golden_weights  # noqa: E402


def main():
    torch.set_num_threads(4)
    tmp = tempfile.mkdtemp()
    cfg = model_kwargs("demo", output_directory=tmp, cnn_hidden_num_channels=4, cnn_kernel_size=3)   # small file
    ref = ReferenceModel(**cfg)
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in golden_weights(cfg, 51).items()}, strict=False)
    ref.eval()
    opt = torch.optim.Adam([p for p in ref.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda t: 0.9 ** (t / 20000.0))
    shape = Shape(batch=4, grid=4, channels=15, input_vocab=14, target_vocab=6, max_command=7, max_target=10, ragged=True)
    for step in range(2):
        b = make_batch(shape, 500 + step)
        logp, _ = ref(commands_input=b["commands"], commands_lengths=b["cmd_lengths"].tolist(),
                      situations_input=b["world"], target_batch=b["targets"], target_lengths=b["tgt_lengths"].tolist())
        ref.get_loss(logp, b["targets"]).backward()
        opt.step(); sched.step(); opt.zero_grad()
        ref.update_state(is_best=(step == 1), accuracy=12.5, exact_match=2.5)
    path = ref.save_checkpoint(file_name="checkpoint.pth.tar", is_best=False, optimizer_state_dict=opt.state_dict())
    out = os.path.join(HERE, "demo_reference_checkpoint.pth.tar")
    shutil.copyfile(path, out)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB; iteration", ref.trained_iterations)

    # reverse direction: this repository's checkpoint into the reference
    from multimodal_seq2seq_gscan_amd.model import Model
    from multimodal_seq2seq_gscan_amd.train import FlatAdam
    ours = Model(**cfg)
    opt_state = ours.load_model(out)
    mine = FlatAdam(ours, 1e-3)
    mine.load_state_dict(opt_state)
    ours_path = ours.save_checkpoint("ours.pth.tar", is_best=False, optimizer_state_dict=mine.state_dict())
    ref2 = ReferenceModel(**cfg)
    opt_state2 = ref2.load_model(ours_path)
    opt2 = torch.optim.Adam([p for p in ref2.parameters() if p.requires_grad], lr=1e-3)
    opt2.load_state_dict(opt_state2)
    for (k, a), (_, b2) in zip(ref.state_dict().items(), ref2.state_dict().items()):
        assert torch.equal(a, b2), k
    assert ref2.trained_iterations == ref.trained_iterations and ref2.best_exact_match == ref.best_exact_match
    s1, s2 = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert all(torch.equal(s1[i]["exp_avg"], s2[i]["exp_avg"]) and torch.equal(s1[i]["exp_avg_sq"], s2[i]["exp_avg_sq"])
               for i in s1)
    print("reverse direction OK: the reference loads this repository's checkpoint (weights, counters, Adam moments)")
    shutil.rmtree(tmp)


def logp_fixture():
    """model.py:228-235 then :206-219 in the reference: what a forward pass must give after loading the checkpoint."""
    import numpy as np
    torch.set_num_threads(4)
    tmp = tempfile.mkdtemp()
    cfg = model_kwargs("demo", output_directory=tmp, cnn_hidden_num_channels=4, cnn_kernel_size=3)
    ref = ReferenceModel(**cfg)
    ref.load_model(os.path.join(HERE, "demo_reference_checkpoint.pth.tar"))
    ref.eval()
    shape = Shape(batch=5, grid=4, channels=15, input_vocab=14, target_vocab=6, max_command=7, max_target=10, ragged=True)
    b = make_batch(shape, 777)
    with torch.no_grad():
        logp, _ = ref(commands_input=b["commands"], commands_lengths=b["cmd_lengths"].tolist(),
                      situations_input=b["world"], target_batch=b["targets"], target_lengths=b["tgt_lengths"].tolist())
        loss = ref.get_loss(logp, b["targets"])
    out = os.path.join(HERE, "demo_reference_checkpoint_logp.npz")
    np.savez_compressed(out, logp=logp.numpy(), loss=np.float32(loss.item()), iteration=np.int64(ref.trained_iterations),
                        **{k: v.numpy() for k, v in b.items()})
    print("wrote", out, os.path.getsize(out), "bytes; loss", float(loss))
    shutil.rmtree(tmp)


if __name__ == "__main__":
    if "--logp-only" in sys.argv:
        logp_fixture()
    else:
        main()
        logp_fixture()
