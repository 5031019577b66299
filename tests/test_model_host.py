"""Host-side behaviour of the drop-in Model that needs no GPU: seeded initialisation identical to the
reference's, checkpoint ABI (state_dict keys), flat parameter views, loud failure on CPU tensors."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN
from multimodal_seq2seq_gscan_amd.config import PARAMETER_TOTALS, model_kwargs
from multimodal_seq2seq_gscan_amd.model import Model


@pytest.fixture(scope="module")
def init_fixture():
    with open(os.path.join(GOLDEN, "init_seed42.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("workload", ["demo", "compositional", "target_length"])
def test_seeded_init_matches_reference(workload, init_fixture):
    """torch.manual_seed(42); Model(**cfg) reproduces the reference's parameters bit for bit
    (train.py:27,58-64): same modules constructed in the same order."""
    torch.manual_seed(42)
    model = Model(**model_kwargs(workload))
    ref = init_fixture[workload]
    assert sum(p.numel() for p in model.parameters()) == PARAMETER_TOTALS[workload] == ref["total"]
    assert list(model.state_dict().keys()) == ref["state_dict_keys"]
    for name, p in model.named_parameters():
        a = p.detach().numpy()
        assert list(a.shape) == ref["params"][name]["shape"], name
        assert hashlib.sha256(a.tobytes()).hexdigest() == ref["params"][name]["sha256"], name


def test_parameters_are_views_of_one_buffer():
    model = Model(**model_kwargs("demo"))
    flat = model.flat_parameters
    assert model.parameter_count == PARAMETER_TOTALS["demo"]
    off = 0
    for name, p in model.named_parameters():       # in order, each on the next 16-byte boundary, zeros in between
        off = (off + 3) // 4 * 4
        assert model._offsets[name] == (off, p.numel())
        assert p.data_ptr() == flat.data_ptr() + 4 * off and p.data_ptr() % 16 == 0
        off += p.numel()
    assert flat.numel() == (off + 3) // 4 * 4
    covered = torch.zeros_like(flat, dtype=torch.bool)
    for o, n in model._offsets.values():
        covered[o:o + n] = True
    assert bool((flat[~covered] == 0).all())
    model.attach_gradients(zero=True)
    for _, p in model.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape
    model.flat_gradients.fill_(2.0)
    assert all(bool((p.grad == 2.0).all()) for p in model.parameters())
    opt = torch.optim.SGD(model.parameters(), lr=0.5)
    before = flat.clone()
    opt.step()                                   # a stock optimizer updates the flat buffer through the views
    assert torch.allclose(flat[covered], before[covered] - 1.0) and bool((flat[~covered] == 0).all())
    opt.zero_grad()                              # set_to_none: views are re-attached (zeroed) on demand
    model.attach_gradients(zero=False)
    assert all(p.grad is not None and bool((p.grad == 0).all()) for p in model.parameters())


def test_checkpoint_roundtrip(tmp_path):
    """Same dictionary keys and file names as model.py:237-261."""
    cfg = model_kwargs("demo", output_directory=str(tmp_path))
    a = Model(**cfg)
    a.update_state(is_best=False)
    a.update_state(is_best=True, accuracy=12.5, exact_match=3.0)
    path = a.save_checkpoint("checkpoint.pth.tar", is_best=True, optimizer_state_dict={"k": 1})
    assert os.path.exists(os.path.join(str(tmp_path), "model_best.pth.tar"))
    ckpt = torch.load(path)
    assert set(ckpt) == {"iteration", "state_dict", "best_iteration", "best_accuracy", "best_exact_match",
                         "optimizer_state_dict"}
    b = Model(**cfg)
    assert b.load_model(path) == {"k": 1}
    assert b.trained_iterations == 2 and b.best_iteration == 2 and b.best_exact_match == 3.0
    assert torch.equal(a.flat_parameters, b.flat_parameters)
    first = next(iter(b.parameters()))
    assert first.data_ptr() == b.flat_parameters.data_ptr()      # loading keeps the views intact


def test_constructor_contract():
    with pytest.raises(ValueError):
        Model(**model_kwargs("demo", attention_type="dot"))       # model.py:95-96
    with pytest.raises(NotImplementedError):
        Model(**model_kwargs("demo", simple_situation_representation=False))
    m = Model(**model_kwargs("demo"), some_unrelated_flag=3, seed=1)   # extras are swallowed (model.py:32)
    assert m.trained_iterations == 0 and m.attention_type == "bahdanau"
    m.update_state(is_best=False)
    assert m.trained_iterations == 1


def test_cpu_tensors_fail_loudly():
    """No CPU fallback in the product path."""
    from multimodal_seq2seq_gscan_amd.synthetic import S0_DEMO, make_batch
    model = Model(**model_kwargs("demo"))
    b = make_batch(S0_DEMO)
    with pytest.raises(RuntimeError, match="HIP device only"):
        model(commands_input=b["commands"], commands_lengths=b["cmd_lengths"].tolist(),
              situations_input=b["world"], target_batch=b["targets"], target_lengths=b["tgt_lengths"].tolist())


def test_reference_checkpoint_loads_and_round_trips(tmp_path):
    """tests/golden/demo_reference_checkpoint.pth.tar was written by the REFERENCE's Model.save_checkpoint after two
    Adam steps (model.py:246-261).  It must load here — weights, counters, and the torch.optim.Adam state into the
    flat fused optimiser — and a checkpoint saved from here must carry the same dictionary (the generating script
    also checks that the reference loads ours)."""
    from multimodal_seq2seq_gscan_amd.train import FlatAdam
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo_reference_checkpoint.pth.tar")
    ref = torch.load(path, map_location="cpu", weights_only=False)
    cfg = model_kwargs("demo", output_directory=str(tmp_path), cnn_hidden_num_channels=4, cnn_kernel_size=3)
    model = Model(**cfg)
    opt_state = model.load_model(path)
    assert model.trained_iterations == ref["iteration"] == 2 and model.best_iteration == ref["best_iteration"] == 2
    assert model.best_exact_match == 2.5 and model.best_accuracy == 12.5
    sd = model.state_dict()
    assert list(sd) == list(ref["state_dict"])
    for k, v in ref["state_dict"].items():
        assert torch.equal(sd[k].cpu(), v), k
    opt = FlatAdam(model, 1e-3)
    opt.load_state_dict(opt_state)
    assert opt.steps_taken == 2
    assert opt.lr_steps == 0 and opt.current_lr() == 1e-3      # the scheduler restarts on resume (train.py:68-84)
    # ... but the FIRST optimizer.step() after the resume runs with the rate the checkpoint carries (the reference's
    # optimizer.load_state_dict overwrites the group's lr behind the new scheduler), the second is back on the curve
    saved_lr = ref["optimizer_state_dict"]["param_groups"][0]["lr"]
    probe = FlatAdam(model, 1e-3)
    probe.load_state_dict(opt_state)
    probe.advance()
    assert probe._lr_of_this_step() == saved_lr and saved_lr < 1e-3
    probe.advance()
    assert abs(probe._lr_of_this_step() - 1e-3 * 0.9 ** (1 / 20000.0)) < 1e-12
    names = [n for n, _ in model.named_parameters()]
    for i, n in enumerate(names):
        off, cnt = model._offsets[n]
        assert torch.equal(opt.exp_avg[off:off + cnt].cpu(), ref["optimizer_state_dict"]["state"][i]["exp_avg"].reshape(-1))
        assert torch.equal(opt.exp_avg_sq[off:off + cnt].cpu(),
                           ref["optimizer_state_dict"]["state"][i]["exp_avg_sq"].reshape(-1))
    out = model.save_checkpoint("again.pth.tar", is_best=False, optimizer_state_dict=opt.state_dict())
    again = torch.load(out, map_location="cpu", weights_only=False)
    assert set(again) == set(ref)
    assert set(again["optimizer_state_dict"]) == set(ref["optimizer_state_dict"])
    g0, r0 = again["optimizer_state_dict"]["param_groups"][0], ref["optimizer_state_dict"]["param_groups"][0]
    assert g0["params"] == r0["params"] and g0["betas"] == tuple(r0["betas"]) and g0["eps"] == r0["eps"]
    for i in ref["optimizer_state_dict"]["state"]:
        a, b = again["optimizer_state_dict"]["state"][i], ref["optimizer_state_dict"]["state"][i]
        assert torch.equal(a["exp_avg"], b["exp_avg"]) and float(a["step"]) == float(b["step"])


def test_drop_in_package_exposes_the_reference_module_names():
    """`seq2seq.{model,train,predict,evaluate,gSCAN_dataset,helpers}` resolve here with the names the reference's
    own callers import (train.py:8-12, predict.py:1-10, __main__.py:5-9)."""
    import importlib
    import torch
    wanted = {"seq2seq.model": ["Model"], "seq2seq.train": ["train"],
              "seq2seq.predict": ["predict", "predict_and_save"], "seq2seq.evaluate": ["evaluate"],
              "seq2seq.gSCAN_dataset": ["GroundedScanDataset", "Vocabulary"],
              "seq2seq.helpers": ["sequence_mask", "log_parameters", "sequence_accuracy"]}
    for module, names in wanted.items():
        m = importlib.import_module(module)
        for n in names:
            assert hasattr(m, n), f"{module}.{n}"
    from seq2seq.helpers import sequence_accuracy, sequence_mask
    assert sequence_mask(torch.tensor([1, 3])).tolist() == [[True, False, False], [True, True, True]]
    assert sequence_mask(torch.tensor([2]), max_len=4).tolist() == [[True, True, False, False]]
    assert sequence_accuracy([1, 2, 3], [1, 2, 3]) == 100 and abs(sequence_accuracy([1, 2, 3], [1, 2]) - 200 / 3) < 1e-9
    assert sequence_accuracy([1], [1, 2]) == 50
