"""Parity of the HIP training step with the reference: golden fixtures (outputs of the reference itself)
and the CPU oracle on the same seeded inputs.  Tolerance 1e-4 absolute on log-probabilities and loss
(BASELINE.json north star, fp32); gradients 1e-4 absolute + 1e-3 relative.  Run: pytest -m gpu."""
import os

import numpy as np
import pytest
import torch

from helpers import fixture_batch, fixture_params, load_fixture
from multimodal_seq2seq_gscan_amd.config import model_kwargs

pytestmark = pytest.mark.gpu
TOL = 1e-4


def build_model(cfg, params):
    from multimodal_seq2seq_gscan_amd.model import Model
    model = Model(**cfg)
    missing, unexpected = model.load_state_dict(params, strict=False)
    assert not unexpected
    assert all("attention_decoder.textual_attention" in k or "attention_decoder.visual_attention" in k for k in missing)
    model = model.cuda()
    model.eval()
    return model


def run_step(model, batch, cfg, weight_target_loss=0.3):
    d = {k: v.cuda() for k, v in batch.items()}
    model.zero_grad()
    logp, aux = model(commands_input=d["commands"], commands_lengths=batch["cmd_lengths"].tolist(),
                      situations_input=d["world"], target_batch=d["targets"],
                      target_lengths=batch["tgt_lengths"].tolist())
    loss = model.get_loss(logp, d["targets"])
    if cfg["auxiliary_task"]:
        loss = loss + weight_target_loss * model.get_auxiliary_loss(aux, d["target_positions"])
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters()}
    return logp.detach().cpu(), aux, loss.item(), grads


def report(name, got, ref):
    err = (got.double() - ref.double()).abs().max().item()
    scale = ref.double().abs().max().item()
    return f"{name}: max|err|={err:.3e} (max|ref|={scale:.3e})", err


def check_against_fixture(name, cfg):
    fx = load_fixture(name)
    params = fixture_params(cfg, fx)
    batch = fixture_batch(fx)
    model = build_model(cfg, params)
    logp, aux, loss, grads = run_step(model, batch, cfg, float(fx["weight_target_loss"]))
    lines, worst = [], 0.0
    msg, err = report("logp", logp, torch.from_numpy(fx["logp"]))
    lines.append(msg)
    assert err < TOL, msg
    assert abs(loss - float(fx["loss"])) < TOL, f"loss {loss} vs {float(fx['loss'])}"
    if cfg["auxiliary_task"]:
        msg, err = report("aux_logp", aux.detach().cpu(), torch.from_numpy(fx["aux_logp"]))
        assert err < TOL, msg
    acc, exact = model.get_metrics(logp.cuda(), batch["targets"].cuda())
    assert abs(acc - float(fx["accuracy"])) < 1e-3 and abs(exact - float(fx["exact_match"])) < 1e-3
    bad = []
    for k, g in grads.items():
        if "grad/" + k in fx:
            ref = torch.from_numpy(fx["grad/" + k])
            msg, err = report(k, g, ref)
            if not torch.allclose(g, ref, atol=TOL, rtol=1e-3):
                bad.append(msg)
        if "gradnorm/" + k in fx:
            ref = float(fx["gradnorm/" + k])
            if abs(g.double().norm().item() - ref) > 1e-3 * max(1.0, ref):
                bad.append(f"{k}: norm {g.double().norm().item():.6e} vs {ref:.6e}")
    assert not bad, "gradient mismatches:\n" + "\n".join(bad)


@pytest.mark.parametrize("cond", [True, False])
@pytest.mark.parametrize("aux", [True, False])
def test_demo_variants(cond, aux):
    """BASELINE config 0 dims (4x4 grid, hidden 20, batch 4, ragged lengths), every head variant."""
    check_against_fixture(f"demo_cond{int(cond)}_aux{int(aux)}.npz",
                          model_kwargs("demo", conditional_attention=cond, auxiliary_task=aux))


@pytest.mark.parametrize("name,overrides", [
    ("demo_enc2.npz", dict(num_encoder_layers=2, auxiliary_task=True)),
    ("demo_enc3_unidirectional.npz", dict(num_encoder_layers=3, conditional_attention=False,
                                          encoder_bidirectional=False)),
])
def test_more_than_one_encoder_layer(name, overrides):
    """nn.LSTM(num_layers=n) in the command encoder (seq2seq_model.py:44-45,76-82), against the reference's own
    outputs: every layer's recurrence, weight gradients and the gradient path between the layers."""
    check_against_fixture(name, model_kwargs("demo", **overrides))


def test_compositional_all_grads():
    """6x6 grid, hidden 100, k=7 (BASELINE config 1 dims) at batch 16, ragged: every gradient tensor."""
    check_against_fixture("compositional_b16.npz", model_kwargs("compositional"))


def test_geca_aux():
    check_against_fixture("geca_aux_b16.npz", model_kwargs("compositional", auxiliary_task=True))


def test_target_length_t120():
    """k=13, T=120: the long-decoder stress configuration (BASELINE config 3 dims)."""
    check_against_fixture("target_length_t120.npz", model_kwargs("target_length"))


def test_intermediates_against_oracle():
    """Every saved activation of the forward pass against the oracle's (localises a failing kernel)."""
    from oracle import seq2seq_oracle as oracle
    cfg = model_kwargs("compositional")
    fx = load_fixture("compositional_b16.npz")
    params, batch = fixture_params(cfg, fx), fixture_batch(fx)
    keep = {}
    oracle.forward(params, batch["commands"], batch["cmd_lengths"], batch["world"], batch["targets"], keep=keep)
    model = build_model(cfg, params)
    d = {k: v.cuda() for k, v in batch.items()}
    with torch.no_grad():
        model(commands_input=d["commands"], commands_lengths=batch["cmd_lengths"].tolist(),
              situations_input=d["world"], target_batch=d["targets"], target_lengths=batch["tgt_lengths"].tolist())
    torch.cuda.synchronize()
    B, L = batch["commands"].shape
    T = batch["targets"].shape[1]
    H, M = cfg["decoder_hidden_size"], 36
    dims = model._dims(B, L, T, 6)
    view = lambda n: model.workspace_view(dims, n).cpu()
    S = view("S").view(B, T, 4 * H)
    stack = lambda k: torch.stack(keep[k], dim=1)
    pairs = {
        "feat": (view("feat").view(B, M, -1), keep["feats"]),
        "enc_out": (view("enc_out").view(B, L, -1), keep["enc_out"]),
        "hN": (view("hN").view(B, -1), keep["hN"]),
        "alpha_c": (view("alpha_c").view(B, T, L), stack("a_c")),
        "alpha_s": (view("alpha_s").view(B, T, M), stack("a_s")),
        "ctx_text": (S[:, :, H:2 * H], stack("ctx_c")),
        "ctx_vis": (S[:, :, 2 * H:3 * H], stack("ctx_s")),
        "h": (S[:, :, 3 * H:], stack("h")),
        "cells": (view("cells").view(B, T, H), stack("c")),
        "gates": (view("gates").view(B, T, 4 * H), stack("gates")),
        "logits": (view("logits").view(B, T, -1), keep["logits"]),
        "att_sum": (view("att_sum").view(B, M), keep["att_sum"]),
    }
    bad = []
    for k, (got, ref) in pairs.items():
        msg, err = report(k, got, ref)
        if not (err < 5e-5):
            bad.append(msg)
    assert not bad, "\n".join(bad)


def test_uint8_world_gives_the_float_world_results_bit_for_bit():
    """The batcher ships the world as uint8 (Grid.encode's dtype, minigrid.py:384): the kernels widen it in registers,
    so log-probabilities and every gradient equal the float32 run's up to the order of the gradient atomics."""
    cfg = model_kwargs("demo", auxiliary_task=True)
    fx = load_fixture("demo_cond1_aux1.npz")
    batch = fixture_batch(fx)
    model = build_model(cfg, fixture_params(cfg, fx))
    logp_f, aux_f, loss_f, grads_f = run_step(model, batch, cfg)
    as_bytes = dict(batch, world=batch["world"].to(torch.uint8))
    assert torch.equal(as_bytes["world"].float(), batch["world"])
    logp_b, aux_b, loss_b, grads_b = run_step(model, as_bytes, cfg)
    assert torch.equal(logp_f, logp_b) and torch.equal(aux_f.detach().cpu(), aux_b.detach().cpu())
    for k in grads_f:
        assert torch.allclose(grads_f[k], grads_b[k], atol=1e-6, rtol=1e-5), k


def test_dropout_host_masks():
    """Train mode with the reference's own CPU-drawn masks handed over (host-mask parity mode)."""
    cfg = model_kwargs("demo")
    fx = load_fixture("demo_dropout_hostmask.npz")
    model = build_model(cfg, fixture_params(cfg, fx))
    model.train()
    batch = fixture_batch(fx)
    d = {k: v.cuda() for k, v in batch.items()}
    B = batch["commands"].shape[0]
    model.set_dropout_masks(torch.from_numpy(fx["mask_cnn"]).reshape(B, 16, -1), torch.from_numpy(fx["mask_enc"]),
                            torch.from_numpy(fx["mask_dec"]))
    logp, _ = model(commands_input=d["commands"], commands_lengths=batch["cmd_lengths"].tolist(),
                    situations_input=d["world"], target_batch=d["targets"], target_lengths=batch["tgt_lengths"].tolist())
    loss = model.get_loss(logp, d["targets"])
    msg, err = report("logp", logp.detach().cpu(), torch.from_numpy(fx["logp"]))
    assert err < TOL, msg
    assert abs(loss.item() - float(fx["loss"])) < TOL


def test_dropout_drawn_inside_the_kernels_against_the_oracle():
    """SURVEY.md 7 hard part 3, production mode: a training-mode step draws its dropout inside the kernels (csrc/dropout.h).
    The masks of its (seed, Philox stream) — written to memory by gscan_dropout_masks_kernel_layout — handed to the CPU oracle
    give the same log-probabilities, loss and gradients: two encoder layers (the inter-layer mask stays in memory), the
    auxiliary head, a ragged batch."""
    from oracle import seq2seq_oracle as oracle
    from multimodal_seq2seq_gscan_amd.model import _dropout_in_kernel
    assert _dropout_in_kernel()
    cfg = model_kwargs("demo", num_encoder_layers=2, auxiliary_task=True)
    fx = load_fixture("demo_enc2.npz")
    params = fixture_params(cfg, fx)
    batch = fixture_batch(fx)
    B, L = batch["commands"].shape
    T, M = batch["targets"].shape[1], batch["world"].shape[1] ** 2
    model = build_model(cfg, params)
    model.train()
    model._dropout_seed = 1234
    calls = model._dropout_calls
    masks = [m.cpu() for m in model._draw_masks(B, L, T, M, torch.device("cuda"), materialize=True)]
    assert len(masks) == 4                                  # cnn, enc, dec and the inter-layer mask
    model._dropout_calls = calls                            # the step below draws the same ones, in its kernels
    ref_loss, ref_grads, ref_logp = oracle.loss_and_grads(params, batch, conditional=True, auxiliary=True, masks=tuple(masks))
    logp, aux, loss, grads = run_step(model, batch, cfg)
    assert model._mask_buffer is None
    assert (logp - ref_logp).abs().max().item() < TOL and abs(loss - ref_loss.item()) < TOL
    for k, g in grads.items():
        assert torch.allclose(g, ref_grads[k], atol=TOL, rtol=1e-3), k


def test_inter_layer_dropout_of_a_two_layer_encoder():
    """nn.LSTM(dropout=p) drops the outputs of every encoder layer but the last: with host masks the HIP step
    matches the oracle (forward and every gradient); with device masks training runs and the loss moves."""
    from oracle import seq2seq_oracle as oracle
    cfg = model_kwargs("demo", num_encoder_layers=2, auxiliary_task=True)
    fx = load_fixture("demo_enc2.npz")
    params = fixture_params(cfg, fx)
    batch = fixture_batch(fx)
    B, L = batch["commands"].shape
    T, He, E, H = batch["targets"].shape[1], cfg["encoder_hidden_size"], cfg["embedding_dimension"], cfg["decoder_hidden_size"]
    gen = torch.Generator().manual_seed(3)
    keep = lambda shape, p: (torch.rand(shape, generator=gen) >= p).float() / (1.0 - p)
    masks = (keep((B, 16, 3 * cfg["cnn_hidden_num_channels"]), 0.1), keep((B, L, E), 0.3), keep((B, T, H), 0.3),
             keep((1, B, L, 2 * He), 0.3))
    ref_loss, ref_grads, ref_logp = oracle.loss_and_grads(params, batch, conditional=True, auxiliary=True, masks=masks)
    model = build_model(cfg, params)
    model.train()
    model.set_dropout_masks(*masks)
    logp, aux, loss, grads = run_step(model, batch, cfg)
    assert (logp - ref_logp).abs().max().item() < TOL and abs(loss - ref_loss.item()) < TOL
    for k, g in grads.items():
        assert torch.allclose(g, ref_grads[k], atol=TOL, rtol=1e-3), k
    # device-drawn masks: two train-mode calls differ, and a few optimiser steps run
    from multimodal_seq2seq_gscan_amd.train import TrainStep
    step = TrainStep(model, learning_rate=1e-2)
    dev = {k: v.cuda() for k, v in batch.items()}
    losses = [float(step(dev)["loss"].item()) for _ in range(8)]
    assert all(l == l for l in losses) and losses[-1] < losses[0]


def test_device_dropout_is_unbiased_and_changes():
    """Production dropout (Philox in HIP): two train-mode calls differ, eval is deterministic."""
    cfg = model_kwargs("demo")
    fx = load_fixture("demo_cond1_aux0.npz")
    model = build_model(cfg, fixture_params(cfg, fx))
    batch = fixture_batch(fx)
    d = {k: v.cuda() for k, v in batch.items()}
    call = lambda: model(commands_input=d["commands"], commands_lengths=batch["cmd_lengths"].tolist(),
                         situations_input=d["world"], target_batch=d["targets"],
                         target_lengths=batch["tgt_lengths"].tolist())[0].detach().cpu()
    e1, e2 = call(), call()
    assert torch.equal(e1, e2)
    model.train()
    t1, t2 = call(), call()
    assert not torch.equal(t1, t2) and torch.isfinite(t1).all()


def test_gradient_accumulation_semantics():
    """backward() ADDS into .grad like autograd does; zero_grad(set_to_none) re-attaches the flat views."""
    cfg = model_kwargs("demo")
    fx = load_fixture("demo_cond1_aux0.npz")
    model = build_model(cfg, fixture_params(cfg, fx))
    batch = fixture_batch(fx)
    _, _, _, g1 = run_step(model, batch, cfg)
    d = {k: v.cuda() for k, v in batch.items()}
    for _ in range(2):   # two more backward passes without zeroing
        logp, _ = model(commands_input=d["commands"], commands_lengths=batch["cmd_lengths"].tolist(),
                        situations_input=d["world"], target_batch=d["targets"],
                        target_lengths=batch["tgt_lengths"].tolist())
        model.get_loss(logp, d["targets"]).backward()
    name = "attention_decoder.lstm.weight_hh_l0"
    g3 = dict(model.named_parameters())[name].grad.cpu()
    assert torch.allclose(g3, 3 * g1[name], atol=1e-5, rtol=1e-4)
    assert dict(model.named_parameters())[name].grad.data_ptr() >= model.flat_gradients.data_ptr()


def test_train_step_matches_reference_adam_steps():
    """TrainStep (fused losses, seeded backward, device-scalar Adam) for three iterations against the reference's
    own Adam + LambdaLR run (tests/golden/demo_adam3.npz)."""
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
    from multimodal_seq2seq_gscan_amd.train import TrainStep
    cfg = model_kwargs("demo", cnn_dropout_p=0.0, encoder_dropout_p=0.0, decoder_dropout_p=0.0)
    fx = load_fixture("demo_adam3.npz")
    shape = Shape(batch=4, grid=4, channels=15, input_vocab=14, target_vocab=6, max_command=7, max_target=10,
                  ragged=True)
    for graph in (False,):
        model = build_model(cfg, fixture_params(cfg, {"seed_weights": 11}))
        step = TrainStep(model, learning_rate=float(fx["lr"]), lr_decay=float(fx["lr_decay"]),
                         lr_decay_steps=float(fx["lr_decay_steps"]))
        for i in range(3):
            batch = {k: v.cuda() for k, v in make_batch(shape, 100 + i).items()}
            if graph:   # one captured shape: pad every batch to the fixture's maximum lengths
                L, T = 7, 10
                batch["commands"] = torch.nn.functional.pad(batch["commands"], (0, L - batch["commands"].shape[1]))
                batch["targets"] = torch.nn.functional.pad(batch["targets"], (0, T - batch["targets"].shape[1]))
            out = step(batch)
            assert abs(out["loss"].item() - float(fx["losses"][i])) < TOL, (graph, i)
        torch.cuda.synchronize()
        for n, p in model.named_parameters():
            assert torch.allclose(p.detach().cpu(), torch.from_numpy(fx["param/" + n]), atol=1e-5, rtol=0), (graph, n)
        assert model.trained_iterations == 3


@pytest.mark.parametrize("auxiliary", [False, True])
def test_fused_loss_backward_matches_two_phase_path(auxiliary):
    """gscan_backward_nll (loss seeded inside the backward kernels, the single-process TrainStep) against the
    statistics -> seeds -> gscan_backward_seeded sequence a data-parallel step uses: same loss, token count and
    gradients (the two differ only in the order the per-row loss terms are summed)."""
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
    from multimodal_seq2seq_gscan_amd.train import TrainStep
    cfg = model_kwargs("demo", auxiliary_task=auxiliary)
    shape = Shape(batch=6, grid=4, channels=15, input_vocab=14, target_vocab=6, max_command=7, max_target=10,
                  ragged=True)
    batch = {k: v.cuda() for k, v in make_batch(shape, 321).items()}
    results = []
    for fused in (False, True):
        model = build_model(cfg, fixture_params(cfg, {"seed_weights": 5}))
        model._dropout_seed = 99
        step = TrainStep(model, learning_rate=0.0, fused_loss=fused)      # lr 0: Adam leaves the parameters alone
        fw = step._section_forward(dict(batch, cmd_lengths=batch["cmd_lengths"].to(torch.int32)))
        step._section_backward(fw)
        torch.cuda.synchronize()
        results.append((step.seeds.cpu().clone(), step.stats.cpu().clone(), model.flat_gradients.cpu().clone()))
    (s0, t0, g0), (s1, t1, g1) = results
    assert torch.allclose(s0, s1, atol=1e-6, rtol=1e-5), (s0, s1)
    assert torch.allclose(t0, t1, atol=1e-4, rtol=1e-5), (t0, t1)
    assert t1[1].item() > 0 and g1.abs().max().item() > 0
    assert torch.allclose(g0, g1, atol=1e-6, rtol=1e-4), (g0 - g1).abs().max().item()


_ONE_CALL_WORKER = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2]); sys.path.insert(0, os.path.join(sys.argv[2], "golden"))
from helpers import fixture_params
from test_parity_gpu import build_model
from multimodal_seq2seq_gscan_amd.config import model_kwargs
from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
from multimodal_seq2seq_gscan_amd.train import TrainStep
aux, sum_red = sys.argv[3] == "1", sys.argv[4] == "1"
cfg = model_kwargs("demo", auxiliary_task=aux)
shape = Shape(batch=6, grid=4, channels=15, input_vocab=14, target_vocab=6, max_command=7, max_target=10, ragged=True)
batch = {k: v.cuda() for k, v in make_batch(shape, 321).items()}
batch = dict(batch, cmd_lengths=batch["cmd_lengths"].to(torch.int32))
model = build_model(cfg, fixture_params(cfg, {"seed_weights": 5}))
model._dropout_seed = 99
step = TrainStep(model, learning_rate=0.0)
stats = torch.zeros(4, device="cuda")
fw = step._section_forward(batch, train_nll=(step.weight_target_loss, sum_red, stats, step.seeds))
torch.cuda.synchronize()
torch.save({"seeds": step.seeds.cpu(), "stats": stats.cpu(), "grads": model.flat_gradients.cpu(), "logp": fw["logp"].cpu()}, sys.argv[5])
"""


@pytest.mark.parametrize("auxiliary,sum_reduction", [(False, False), (True, False), (False, True)])
def test_one_call_train_step_equals_forward_then_backward(tmp_path, auxiliary, sum_reduction):
    """gscan_train_step_nll (forward + loss + backward in one library call, what TrainStep issues) against
    gscan_forward followed by gscan_backward_nll, in a child process: same log-probabilities, loss, statistics and
    gradients."""
    import os, subprocess, sys
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
    from multimodal_seq2seq_gscan_amd.train import TrainStep
    cfg = model_kwargs("demo", auxiliary_task=auxiliary)
    shape = Shape(batch=6, grid=4, channels=15, input_vocab=14, target_vocab=6, max_command=7, max_target=10, ragged=True)
    batch = {k: v.cuda() for k, v in make_batch(shape, 321).items()}
    batch = dict(batch, cmd_lengths=batch["cmd_lengths"].to(torch.int32))
    model = build_model(cfg, fixture_params(cfg, {"seed_weights": 5}))
    model._dropout_seed = 99
    step = TrainStep(model, learning_rate=0.0)
    stats = torch.zeros(4, device="cuda")
    fw = step._section_forward(batch)
    model._launch_backward_nll(fw["call"], step.weight_target_loss, stats, step.seeds, sum_reduction=sum_reduction)
    torch.cuda.synchronize()
    ref = {"seeds": step.seeds.cpu().clone(), "stats": stats.cpu().clone(), "grads": model.flat_gradients.cpu().clone(),
           "logp": fw["logp"].cpu().clone()}
    here = os.path.dirname(os.path.abspath(__file__))
    for fused_decoder in ("0",):
        out = str(tmp_path / f"one_call_{fused_decoder}.pt")
        env = dict(os.environ)
        r = subprocess.run([sys.executable, "-c", _ONE_CALL_WORKER, os.path.dirname(here), here, str(int(auxiliary)),
                            str(int(sum_reduction)), out], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        got = torch.load(out)
        assert torch.equal(got["logp"], ref["logp"]), fused_decoder                  # the forward pass is bitwise the same
        assert torch.allclose(got["seeds"], ref["seeds"], atol=1e-6, rtol=1e-5), (fused_decoder, got["seeds"], ref["seeds"])
        assert torch.allclose(got["stats"], ref["stats"], atol=1e-4, rtol=1e-5), (fused_decoder, got["stats"], ref["stats"])
        assert got["stats"][1].item() > 0 and got["grads"].abs().max().item() > 0
        assert torch.allclose(got["grads"], ref["grads"], atol=1e-6, rtol=1e-4), (fused_decoder, (got["grads"] - ref["grads"]).abs().max())


def test_decode_input_steps_match_teacher_forced_forward():
    """encode_input + one decode_input per target token (the greedy-decoding surface, predict.py:82-106) reproduces
    the teacher-forced forward pass: same logits step by step, same attention, same final state."""
    cfg = model_kwargs("demo", conditional_attention=True, auxiliary_task=True)
    fx = load_fixture("demo_cond1_aux1.npz")
    model = build_model(cfg, fixture_params(cfg, fx)).eval()
    batch = fixture_batch(fx)
    d = {k: v.cuda() for k, v in batch.items()}
    with torch.no_grad():
        logp, _ = model(commands_input=d["commands"], commands_lengths=batch["cmd_lengths"].tolist(),
                        situations_input=d["world"], target_batch=d["targets"],
                        target_lengths=batch["tgt_lengths"].tolist())
        enc = model.encode_input(commands_input=d["commands"], commands_lengths=batch["cmd_lengths"].tolist(),
                                 situations_input=d["world"])
        keys_vis = model.visual_attention.key_layer(enc["encoded_situations"])
        keys_txt = model.textual_attention.key_layer(enc["encoded_commands"]["encoder_outputs"])
        hidden = model.attention_decoder.initialize_hidden(
            model.tanh(model.enc_hidden_to_dec_hidden(enc["hidden_states"])))
        B, T = d["targets"].shape
        att = torch.zeros(B, keys_vis.shape[1], device="cuda")
        for t in range(T):
            out, hidden, ctx, a_txt, a_vis = model.decode_input(
                target_token=d["targets"][:, t], hidden=hidden, encoder_outputs=keys_txt,
                input_lengths=batch["cmd_lengths"].tolist(), encoded_situations=keys_vis)
            assert torch.allclose(torch.log_softmax(out, -1), logp[:, t], atol=TOL), t
            assert torch.allclose(a_vis.sum(1), torch.ones(B, device="cuda"), atol=1e-5)
            att += a_vis
        aux = model.auxiliary_task_forward(att)
    assert torch.allclose(logp.cpu(), torch.from_numpy(fx["logp"]), atol=TOL)
    assert torch.allclose(aux.cpu(), torch.from_numpy(fx["aux_logp"]), atol=TOL)
    # the encoder side of the dictionary against the oracle
    from oracle import seq2seq_oracle as oracle
    params = fixture_params(cfg, fx)
    feats = oracle.world_encoder(params, batch["world"])
    hN, enc_out = oracle.command_encoder(params, batch["commands"], batch["cmd_lengths"])
    assert torch.allclose(enc["encoded_situations"].cpu(), feats, atol=5e-5)
    assert torch.allclose(enc["encoded_commands"]["encoder_outputs"].cpu(), enc_out.transpose(0, 1), atol=5e-5)
    assert torch.allclose(enc["hidden_states"].cpu(), hN, atol=5e-5)


def test_decode_input_batched_from_handed_in_encodings_equals_forward():
    """Model.decode_input_batched (model.py:190-204) fed with the dictionary of encode_input gives forward()'s
    log-probabilities (time-major, as the reference returns them) and the summed visual attention behind the
    auxiliary head; encodings that were modified by the caller are honoured."""
    cfg = model_kwargs("demo", auxiliary_task=True)
    fx = load_fixture("demo_cond1_aux1.npz")
    model = build_model(cfg, fixture_params(cfg, fx))
    batch = fixture_batch(fx)
    d = {k: v.cuda() for k, v in batch.items()}
    lens = batch["cmd_lengths"].tolist()
    with torch.no_grad():
        logp, aux = model(commands_input=d["commands"], commands_lengths=lens, situations_input=d["world"],
                          target_batch=d["targets"], target_lengths=batch["tgt_lengths"].tolist())
        enc = model.encode_input(commands_input=d["commands"], commands_lengths=lens, situations_input=d["world"])
        out, att = model.decode_input_batched(
            target_batch=d["targets"], target_lengths=batch["tgt_lengths"].tolist(), initial_hidden=enc["hidden_states"],
            encoded_commands=enc["encoded_commands"]["encoder_outputs"], command_lengths=lens,
            encoded_situations=enc["encoded_situations"])
        assert tuple(out.shape) == (logp.shape[1], logp.shape[0], logp.shape[2])
        assert (out.transpose(0, 1) - logp).abs().max().item() < 1e-6
        assert (model.auxiliary_task_forward(att) - aux).abs().max().item() < 1e-5
        assert (out.transpose(0, 1).cpu() - torch.from_numpy(fx["logp"])).abs().max().item() < TOL
        other, _ = model.decode_input_batched(
            target_batch=d["targets"], target_lengths=batch["tgt_lengths"].tolist(),
            initial_hidden=enc["hidden_states"] * 0.5, encoded_commands=enc["encoded_commands"]["encoder_outputs"],
            command_lengths=lens, encoded_situations=enc["encoded_situations"])
        assert (other - out).abs().max().item() > 1e-3


def test_greedy_predict_matches_reference_loop():
    """predict() (batched greedy decoding on the HIP path) against the reference's own per-example
    encode_input / decode_input loop (tests/golden/demo_greedy.npz): same tokens, stopping steps, attention."""
    from multimodal_seq2seq_gscan_amd.predict import evaluate, predict
    fx = load_fixture("demo_greedy.npz")
    cfg = model_kwargs("demo", conditional_attention=True, auxiliary_task=True)
    model = build_model(cfg, fixture_params(cfg, fx))
    batch = fixture_batch(fx)
    d = {k: v.cuda() for k, v in batch.items()}
    B = d["commands"].shape[0]

    def iterator(batch_size):
        for lo in range(0, B, batch_size):
            hi = min(B, lo + batch_size)
            yield (d["commands"][lo:hi], batch["cmd_lengths"][lo:hi].tolist(), [f"d{i}" for i in range(lo, hi)],
                   d["world"][lo:hi], [{"id": i} for i in range(lo, hi)], d["targets"][lo:hi],
                   batch["tgt_lengths"][lo:hi].tolist(), None, d["target_positions"][lo:hi])

    sos, eos, steps = int(fx["sos"]), int(fx["eos"]), int(fx["max_steps"])
    for batch_size in (B, 1, 4):
        outs = list(predict(iterator(batch_size), model, steps, 0, sos, eos))
        assert len(outs) == B
        for r, (inp, deriv, spec, out_seq, tgt, a_txt, a_vis, aux_acc) in enumerate(outs):
            n = int(fx["nsteps"][r])
            ref_tokens = fx["tokens"][r, :n].tolist()
            if ref_tokens[-1] == eos:
                ref_tokens, n = ref_tokens[:-1], n - 1
            assert out_seq == ref_tokens, (batch_size, r)
            assert deriv == f"d{r}" and spec == {"id": r}
            L = int(batch["cmd_lengths"][r])
            assert inp.shape == (1, L) and len(a_txt) == n and len(a_vis) == n
            if n:
                assert torch.allclose(torch.tensor(a_txt), torch.from_numpy(fx["alpha_text"][r, :n, :L]), atol=1e-4)
                assert torch.allclose(torch.tensor(a_vis), torch.from_numpy(fx["alpha_vis"][r, :n]), atol=1e-4)
    acc, exact, aux = evaluate(iterator(3), model, steps, 0, sos, eos)
    assert 0.0 <= acc <= 100.0 and 0.0 <= exact <= 100.0 and 0.0 <= aux <= 100.0


def test_persistent_greedy_decoder_equals_the_stepwise_loop_and_stops_at_the_limit():
    """One launch per batch (gscan_greedy_decode: argmax and <EOS> test fed back in-kernel) against the reference's
    own call sequence driven token by token (encode_input / decode_input, predict.py:82-112): same tokens, stopping
    steps, attention rows and summed visual attention, at the paper's dims for a ragged batch; and the T = 120 limit
    of the target-length configuration (max_decoding_steps = 119: rows that never emit <EOS> stop after exactly 120
    steps, predict.py:101)."""
    from multimodal_seq2seq_gscan_amd.model import Model
    from multimodal_seq2seq_gscan_amd.predict import greedy_decode, greedy_decode_stepwise
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
    # (with seeded random weights every row of a batch behaves alike: which token plays <EOS> decides whether the
    # rows stop early or run into the limit; the CPU oracle's greedy_decode gives the same lengths for these seeds)
    for workload, limit, uint8, eos, expect in (("compositional", 12, False, 5, "early"), ("compositional", 12, True, 2, "limit"),
                                                ("target_length", 119, True, 6, "early"), ("target_length", 119, False, 2, "limit")):
        torch.manual_seed(5)
        cfg = model_kwargs(workload, auxiliary_task=True)
        model = Model(**cfg).cuda().eval()
        batch = make_batch(Shape(batch=37, input_vocab=cfg["input_vocabulary_size"],
                                 target_vocab=cfg["target_vocabulary_size"], ragged=True), seed=21)
        world = batch["world"].to(torch.uint8).cuda() if uint8 else batch["world"].cuda()
        args = (model, batch["commands"].cuda(), batch["cmd_lengths"].tolist(), world, 1, eos, limit)
        with torch.no_grad():
            one, ref = greedy_decode(*args), greedy_decode_stepwise(*args)
        assert one["tokens"] == ref["tokens"]
        lengths = [len(t) for t in one["tokens"]]
        if expect == "limit":
            assert lengths == [limit + 1] * 37 and all(eos not in t for t in one["tokens"])
        else:
            assert max(lengths) <= limit and all(t[-1] == eos and eos not in t[:-1] for t in one["tokens"])
        for r in range(37):
            assert torch.allclose(torch.tensor(one["alpha_text"][r]), torch.tensor(ref["alpha_text"][r]), atol=1e-5)
            assert torch.allclose(torch.tensor(one["alpha_vis"][r]), torch.tensor(ref["alpha_vis"][r]), atol=1e-5)
        assert torch.allclose(one["att_sum"], ref["att_sum"], atol=1e-4)


@pytest.mark.parametrize("overrides,shape_kw", [
    ({"encoder_hidden_size": 128, "decoder_hidden_size": 128, "embedding_dimension": 25}, dict(grid=6)),
    ({"encoder_hidden_size": 100, "decoder_hidden_size": 100, "embedding_dimension": 25, "conditional_attention": False},
     dict(grid=12, max_command=90)),
    ({"encoder_hidden_size": 30, "decoder_hidden_size": 50, "embedding_dimension": 5, "encoder_bidirectional": False},
     dict(grid=4)),
], ids=["hidden128", "grid12_command90", "hidden50_encoder30"])
def test_greedy_decoding_on_streamed_shapes_against_oracle(overrides, shape_kw):
    """predict.py:57-128 on shapes only the streaming kernels take: the one-launch greedy decoder and the stepwise
    decode_input loop against the oracle's per-row loop — same tokens, same attention rows."""
    from multimodal_seq2seq_gscan_amd.predict import greedy_decode, greedy_decode_stepwise
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
    from oracle import seq2seq_oracle as oracle
    from weights import golden_weights
    cfg = model_kwargs("demo", cnn_dropout_p=0.0, encoder_dropout_p=0.0, decoder_dropout_p=0.0, **overrides)
    kw = dict(batch=5, grid=4, channels=15, input_vocab=14, target_vocab=6, max_command=7, max_target=10, ragged=True)
    kw.update(shape_kw)
    batch = make_batch(Shape(**kw), seed=11)
    params = {k: torch.from_numpy(v) for k, v in golden_weights(cfg, 23).items()}
    model = build_model(cfg, params).eval()
    sos, limit = 1, 9
    for eos in (2, 4):
        args = (model, batch["commands"].cuda(), batch["cmd_lengths"].tolist(), batch["world"].cuda(), sos, eos, limit)
        with torch.no_grad():
            one, step = greedy_decode(*args), greedy_decode_stepwise(*args)
        ref = oracle.greedy_decode(params, batch["commands"], batch["cmd_lengths"], batch["world"], sos, eos, limit,
                                   conditional=cfg["conditional_attention"], bidirectional=cfg["encoder_bidirectional"])
        for r, row in enumerate(ref):
            assert one["tokens"][r] == row["tokens"] == step["tokens"][r], (eos, r)
            n, L = len(row["tokens"]), int(batch["cmd_lengths"][r])
            a_t = torch.stack([a.flatten()[:L] for a in row["alpha_text"]])
            a_v = torch.stack([a.flatten() for a in row["alpha_vis"]])
            assert torch.allclose(torch.tensor(one["alpha_text"][r])[:n, :L], a_t, atol=1e-4)
            assert torch.allclose(torch.tensor(one["alpha_vis"][r])[:n], a_v, atol=1e-4)


@pytest.mark.parametrize("gate_images", ["0", "1"])
def test_fixture_suite_on_the_streaming_kernels(gate_images):
    """GSCAN_DECODER_ANY=1 GSCAN_ENCODER_ANY=1 (read once per process, hence the child process): the reference's own
    outputs — training step fixtures, the greedy fixture, the decode_input sequences — through the streaming kernels
    instead of the register/LDS-resident ones, in both of their forms (GSCAN_ANY_U: W_ih's context columns streamed, or
    the gate images U read in their place as the resident kernels do)."""
    import subprocess
    import sys
    env = dict(os.environ, GSCAN_DECODER_ANY="1", GSCAN_ENCODER_ANY="1", GSCAN_ANY_U=gate_images)
    pick = ("demo_variants or more_than_one_encoder_layer or geca_aux or greedy_predict_matches or decode_input or "
            "persistent_greedy or one_call_train_step")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", pick,
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_fixture_suite_in_deterministic_mode():
    """GSCAN_DETERMINISTIC=1 (read once per process, hence the child process): split-K partial tiles added in slice order,
    embedding and convolution-bias gradients by their ordered kernels — the reference's own outputs through that path."""
    import subprocess
    import sys
    env = dict(os.environ, GSCAN_DETERMINISTIC="1")
    pick = ("demo_variants or more_than_one_encoder_layer or geca_aux or compositional_all_grads or target_length_t120 or "
            "one_call_train_step or train_step_matches_reference_adam")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", pick,
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.parametrize("extra_env", [{}, {"GSCAN_DECODER_ANY": "1", "GSCAN_ENCODER_ANY": "1"}, {"GSCAN_DETERMINISTIC": "1"}],
                         ids=["resident_kernels", "streaming_kernels", "deterministic"])
def test_fixture_suite_with_attention_gradients_summed_over_time_first(extra_env):
    """GSCAN_TIME_REDUCED_T=1 (read once per process, hence the child process): the backward pass of long target sequences —
    G = alpha^T . [delta | dzq] per memory, then d PK += G . W and the context columns of dW_ih / dW_q2k as G^T . PK
    (csrc/step.hip attention_time_reduced, csrc/attention_grad.hip alpha_reduce_kernel) — forced on EVERY sequence length:
    the reference's own training-step outputs and gradients through that path (seq2seq_model.py:414-425 and their autograd
    duals, model.py:190-219)."""
    import subprocess
    import sys
    env = dict(os.environ, GSCAN_TIME_REDUCED_T="1", **extra_env)
    pick = ("demo_variants or more_than_one_encoder_layer or geca_aux or compositional_all_grads or target_length_t120 or "
            "one_call_train_step or train_step_matches_reference_adam or edge_shapes")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", pick,
                        "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


def test_command_line_train_then_test_modes(tmp_path):
    """`python -m seq2seq --mode=train ... --synthetic_data` then `--mode=test` (seq2seq/__main__.py:21-167): the
    training loop runs, writes the reference's checkpoint dictionary, and the test mode decodes greedily from it
    and writes <split>_predict.json with the reference's record schema (predict.py:44-51)."""
    import json
    from multimodal_seq2seq_gscan_amd.__main__ import main, parser
    out = str(tmp_path)
    common = ["--output_directory", out, "--synthetic_data", "--training_batch_size", "8", "--seed", "3"]
    main(vars(parser.parse_args(["--mode", "train", "--max_training_iterations", "3", "--synthetic_batches", "3",
                                 "--print_every", "2"] + common)))
    ckpt = torch.load(f"{out}/checkpoint.pth.tar", map_location="cpu", weights_only=False)
    assert ckpt["iteration"] == 3 and "state_dict" in ckpt and "optimizer_state_dict" in ckpt
    main(vars(parser.parse_args(["--mode", "test", "--resume_from_file", f"{out}/checkpoint.pth.tar",
                                 "--max_testing_examples", "5", "--max_decoding_steps", "6", "--splits", "test,dev"]
                                + common)))
    for split in ("test", "dev"):
        records = json.load(open(f"{out}/{split}_predict.json"))
        assert len(records) == 5
        for rec in records:
            assert set(rec) == {"input", "prediction", "derivation", "target", "situation", "attention_weights_input",
                                "attention_weights_situation", "accuracy", "exact_match", "position_accuracy"}
            assert len(rec["prediction"]) <= 7 and len(rec["attention_weights_situation"]) == len(rec["prediction"])


def test_sum_reduction_backward_plus_mean_adam_equals_the_fused_step():
    """The data-parallel building blocks on one GPU: gscan_backward_nll(sum_reduction=1) writes sum-loss gradients
    and the statistics behind them; gscan_adam_step_mean divides by the token count.  Gradients / count must equal
    the mean-loss gradients of the single-process fused step, and the update must be Adam's on those."""
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
    from multimodal_seq2seq_gscan_amd.train import TrainStep
    from oracle import seq2seq_oracle as oracle
    cfg = model_kwargs("demo", cnn_dropout_p=0.0, encoder_dropout_p=0.0, decoder_dropout_p=0.0)
    shape = Shape(batch=6, grid=4, channels=15, input_vocab=14, target_vocab=6, max_command=7, max_target=10,
                  ragged=True)
    batch = {k: v.cuda() for k, v in make_batch(shape, 55).items()}
    batch["cmd_lengths"] = batch["cmd_lengths"].to(torch.int32)
    a = build_model(cfg, fixture_params(cfg, {"seed_weights": 8}))
    b = build_model(cfg, fixture_params(cfg, {"seed_weights": 8}))
    a.train(), b.train()
    sa, sb = TrainStep(a, learning_rate=1e-2), TrainStep(b, learning_rate=1e-2)
    sa._host_prologue(), sb._host_prologue()
    sa._section_backward(sa._section_forward(batch))            # mean loss (gscan_backward_nll, sum_reduction=0)
    fw = sb._section_forward(batch)
    store = b._grad_store
    b._launch_backward_nll(fw["call"], sb.weight_target_loss, store[-4:], sb.seeds, sum_reduction=True)
    _, count, loss = sb.exchange.mean_from_sums(store)
    torch.cuda.synchronize()
    assert abs(loss.item() - sa.seeds[2].item()) < 1e-6 and count.item() == sa.stats[1].item()
    g_mean = (b.flat_gradients / count).cpu()
    assert torch.allclose(g_mean, a.flat_gradients.cpu(), atol=1e-7, rtol=1e-4)
    # the optimiser launch: Adam on grad / count, gradients cleared
    p0 = b.flat_parameters.cpu().clone()
    sb.optimizer.launch_mean(count)
    torch.cuda.synchronize()
    ref = [p0.clone()]
    oracle.adam_step(ref, [g_mean], [torch.zeros_like(p0)], [torch.zeros_like(p0)], 1, 1e-2, lr_decay=0.9,
                     lr_decay_steps=20000.0)
    solid = g_mean.abs() > 1e-5            # Adam's first step is lr * g / (|g| + eps): rounding noise in |g| ~ eps
    got = b.flat_parameters.cpu()          # is amplified to the size of the step, so compare where g is solid
    assert solid.sum().item() > 0.5 * solid.numel()
    assert torch.allclose(got[solid], ref[0][solid], atol=2e-6, rtol=0)
    assert (got - p0).abs().max().item() <= 1e-2 * 1.001
    assert b.flat_gradients.abs().max().item() == 0.0          # zero_grad folded into the optimiser launch


def test_one_collective_train_step_tracks_the_fused_step():
    """TrainStep's data-parallel form (sum-loss backward, statistics behind the gradients, mean Adam), run with a
    single process where the all-reduce is the identity: three steps give the losses of the single-process fused
    step and parameters within the optimiser's step size of it (Adam amplifies rounding noise in ~1e-9 gradients
    to the size of one step, so the comparison cannot be tighter than that)."""
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
    from multimodal_seq2seq_gscan_amd.train import TrainStep
    cfg = model_kwargs("demo", cnn_dropout_p=0.0, encoder_dropout_p=0.0, decoder_dropout_p=0.0)
    shape = Shape(batch=6, grid=4, channels=15, input_vocab=14, target_vocab=6, max_command=7, max_target=10,
                  ragged=True)
    a = build_model(cfg, fixture_params(cfg, {"seed_weights": 9}))
    b = build_model(cfg, fixture_params(cfg, {"seed_weights": 9}))
    sa = TrainStep(a, learning_rate=1e-3)
    sb = TrainStep(b, learning_rate=1e-3, single_exchange=True)
    assert sa.fused_loss and sb.single_exchange and not sb.fused_loss
    for i in range(3):
        batch = {k: v.cuda() for k, v in make_batch(shape, 300 + i).items()}
        la, lb = sa(batch)["loss"].item(), sb(batch)["loss"].item()
        assert abs(la - lb) < 1e-4, (i, la, lb)
    torch.cuda.synchronize()
    assert (a.flat_parameters - b.flat_parameters).abs().max().item() < 3.5e-3
    differing = ((a.flat_parameters - b.flat_parameters).abs() > 1e-5).float().mean().item()
    assert differing < 0.05, differing
    assert a.trained_iterations == b.trained_iterations == 3


EDGE_CASES = [
    # name, model overrides, Shape overrides, post-processing of the batch
    ("single_row", {}, dict(batch=1), None),
    ("odd_batch_min_lengths", {}, dict(batch=3, max_command=3, max_target=2), None),
    ("one_step_targets", {}, dict(batch=2, max_target=1), "sos_only"),
    ("t33_crosses_head_chunk", {}, dict(batch=2, max_target=33), None),
    ("t65_three_chunks", {"conditional_attention": False}, dict(batch=2, max_target=65), None),
    ("grid3", {}, dict(batch=2, grid=3), None),
    ("grid8_max_cells_aux", {"auxiliary_task": True}, dict(batch=2, grid=8), None),
    ("hidden32_unidirectional", {"encoder_hidden_size": 32, "decoder_hidden_size": 32, "encoder_bidirectional": False,
                                 "embedding_dimension": 6}, dict(batch=3), None),
    ("hidden64_k3_nocond", {"encoder_hidden_size": 64, "decoder_hidden_size": 64, "cnn_kernel_size": 3,
                            "conditional_attention": False, "embedding_dimension": 8}, dict(batch=2), None),
    ("length_one_commands", {}, dict(batch=4), "short_commands"),
    # hidden sizes that got kernels in round 2 (every multiple of 4: decoder up to 100, encoder up to 128)
    ("hidden24_encoder72", {"encoder_hidden_size": 72, "decoder_hidden_size": 24, "embedding_dimension": 7}, dict(batch=3), None),
    ("hidden72_encoder128_aux", {"encoder_hidden_size": 128, "decoder_hidden_size": 72, "auxiliary_task": True},
     dict(batch=2), None),
    # two bidirectional layers at a small hidden size: both directions add their input gradients into one buffer from
    # one launch with K = 4 He too short to split (they raced before the products were made atomic)
    ("enc2_bidirectional_hidden8", {"encoder_hidden_size": 8, "num_encoder_layers": 2}, dict(batch=2, max_command=2), None),
    ("enc3_bidirectional_hidden12", {"encoder_hidden_size": 12, "num_encoder_layers": 3}, dict(batch=3), None),
    ("hidden4", {"encoder_hidden_size": 4, "decoder_hidden_size": 4, "embedding_dimension": 3}, dict(batch=2), None),
    # 8x8 grid at hidden 100: the visual gate images (102 KB per row) do not fit LDS next to the rest and are
    # streamed from L2 by the decoder kernels (round 1 refused this shape)
    ("grid8_hidden100_streamed_gate_images", {"encoder_hidden_size": 100, "decoder_hidden_size": 100,
                                              "embedding_dimension": 25, "auxiliary_task": True},
     dict(batch=3, grid=8, max_target=6), None),
    ("all_pad_targets_row", {}, dict(batch=3), "pad_row"),
    # round 4 — shapes the reference takes that the register/LDS-resident kernels have no variant for: they run on the
    # streaming kernels (decoder_any.hip, encoder_lstm_*_any_kernel, keys_backward_any_kernel)
    ("hidden128_decoder_streams", {"encoder_hidden_size": 128, "decoder_hidden_size": 128, "embedding_dimension": 25},
     dict(batch=3, max_target=6), None),
    ("hidden200_aux", {"encoder_hidden_size": 200, "decoder_hidden_size": 200, "embedding_dimension": 25,
                       "auxiliary_task": True}, dict(batch=2, grid=6, max_target=5), None),
    ("hidden256_nocond_unidirectional", {"encoder_hidden_size": 256, "decoder_hidden_size": 256, "embedding_dimension": 32,
                                         "conditional_attention": False, "encoder_bidirectional": False},
     dict(batch=2, max_target=4), None),
    # round 5 — decoder hidden sizes / embedding widths above 256 (the embedding gradients walk column blocks of 256)
    ("hidden320_embedding300", {"encoder_hidden_size": 100, "decoder_hidden_size": 320, "embedding_dimension": 300},
     dict(batch=2, max_target=4), None),
    ("hidden513_odd_aux", {"encoder_hidden_size": 36, "decoder_hidden_size": 513, "embedding_dimension": 7,
                           "auxiliary_task": True}, dict(batch=2, grid=4, max_target=3), None),
    ("hidden50_encoder30_not_multiples_of_4", {"encoder_hidden_size": 30, "decoder_hidden_size": 50, "embedding_dimension": 5},
     dict(batch=3), None),
    ("grid12_144_cells_aux", {"encoder_hidden_size": 100, "decoder_hidden_size": 100, "embedding_dimension": 25,
                              "auxiliary_task": True}, dict(batch=2, grid=12, max_target=5), None),
    ("command_of_100_tokens", {"encoder_hidden_size": 100, "decoder_hidden_size": 100, "embedding_dimension": 25},
     dict(batch=3, max_command=100, max_target=5), None),
    ("command_of_128_tokens_two_layers_hidden256", {"encoder_hidden_size": 256, "decoder_hidden_size": 64,
                                                     "embedding_dimension": 16, "num_encoder_layers": 2},
     dict(batch=2, max_command=128, max_target=4), None),
    ("grid10_command70_hidden20", {}, dict(batch=3, grid=10, max_command=70), None),
]


@pytest.mark.parametrize("name,overrides,shape_kw,post", EDGE_CASES, ids=[c[0] for c in EDGE_CASES])
def test_edge_shapes_against_oracle(name, overrides, shape_kw, post):
    """Ragged / minimal / maximal shapes and the other compiled hidden sizes: log-probabilities, loss and every
    gradient of the HIP step against the CPU oracle (itself pinned to the reference) on the same seeded inputs."""
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
    from oracle import seq2seq_oracle as oracle
    from weights import golden_weights
    cfg = model_kwargs("demo", cnn_dropout_p=0.0, encoder_dropout_p=0.0, decoder_dropout_p=0.0, **overrides)
    kw = dict(batch=4, grid=4, channels=15, input_vocab=14, target_vocab=6, max_command=7, max_target=10, ragged=True)
    kw.update(shape_kw)
    batch = make_batch(Shape(**kw), seed=len(name))
    if post == "sos_only":                 # T = 1: only the SOS column (every shifted target is the literal 0 = pad)
        batch["targets"] = batch["targets"][:, :1].contiguous()
        batch["tgt_lengths"] = torch.ones_like(batch["tgt_lengths"])
    elif post == "short_commands":         # rows with a single real token
        batch["cmd_lengths"] = torch.tensor([batch["commands"].shape[1], 1, 1, 2])
        for r, n in enumerate(batch["cmd_lengths"].tolist()):
            batch["commands"][r, n:] = 0
    elif post == "pad_row":                # a row whose targets are all padding contributes nothing to the loss
        batch["targets"][1, 1:] = 0
        batch["tgt_lengths"][1] = 1
    params = {k: torch.from_numpy(v) for k, v in golden_weights(cfg, 17).items()}
    model = build_model(cfg, params)
    logp, aux, loss, grads = run_step(model, batch, cfg)
    ref_loss, ref_grads, ref_logp = oracle.loss_and_grads(params, batch, conditional=cfg["conditional_attention"],
                                                          auxiliary=cfg["auxiliary_task"],
                                                          bidirectional=cfg["encoder_bidirectional"])
    assert torch.isfinite(logp).all()
    assert (logp - ref_logp).abs().max().item() < TOL, name
    if torch.isfinite(ref_loss):
        assert abs(loss - ref_loss.item()) < TOL, (loss, ref_loss.item())
        for k, g in grads.items():
            assert torch.allclose(g, ref_grads[k], atol=TOL, rtol=1e-3), (name, k, (g - ref_grads[k]).abs().max().item())


def test_command_line_on_a_dataset_file(tmp_path):
    """`python -m seq2seq --mode=train --data_directory=... --generate_vocabularies` on a gSCAN dataset file (the
    README's demo command, README.md:177, on the README's published example and variations of it), with evaluation
    and best-checkpointing (train.py:129-149), then `--mode=test` writing <split>_predict.json with words."""
    import json
    import shutil
    from multimodal_seq2seq_gscan_amd.__main__ import main, parser
    data_dir = tmp_path / "data"
    data_dir.mkdir()
    shutil.copyfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "readme_demo_dataset.txt"),
                    data_dir / "dataset.txt")
    out = str(tmp_path / "out")
    common = ["--data_directory", str(data_dir), "--output_directory", out, "--embedding_dimension", "5",
              "--encoder_hidden_size", "20", "--decoder_hidden_size", "20", "--seed", "1"]
    main(vars(parser.parse_args(["--mode", "train", "--max_training_iterations", "60", "--training_batch_size", "5",
                                 "--print_every", "20", "--evaluate_every", "30", "--generate_vocabularies",
                                 "--max_decoding_steps", "8", "--learning_rate", "0.01"] + common)))
    assert os.path.exists(data_dir / "training_input_vocab.txt") and os.path.exists(data_dir / "training_target_vocab.txt")
    ckpt = torch.load(f"{out}/model_best.pth.tar", map_location="cpu", weights_only=False)
    assert ckpt["best_exact_match"] > 0 and ckpt["state_dict"]["encoder.embedding.weight"].shape == (10, 5)
    main(vars(parser.parse_args(["--mode", "test", "--resume_from_file", f"{out}/model_best.pth.tar",
                                 "--splits", "test,situational_1", "--max_decoding_steps", "8"] + common)))
    records = json.load(open(f"{out}/situational_1_predict.json"))
    assert len(records) == 1 and records[0]["input"] == ["walk", "to", "a", "red", "circle"]
    assert records[0]["target"] == ["turn left", "turn left", "walk", "turn left", "walk"]
    assert all(w in ("turn left", "turn right", "walk", "<EOS>", "<SOS>", "<PAD>") for w in records[0]["prediction"])
    assert len(json.load(open(f"{out}/test_predict.json"))) == 2


def test_staged_batcher_delivers_the_packed_rows_while_the_step_runs(tmp_path):
    """batches gathered into a ring of pinned slabs, copied a batch ahead on a copy stream, world kept
    uint8.  Over several epochs (the ring of three slabs is reused many times, with training steps in flight on the
    delivered views) every delivered tensor equals the packed rows it was cut from; the reference-contract iterator
    on the device equals the host one; and training on staged uint8 batches gives the loss of training on the
    same rows shipped as float32 tensors."""
    from multimodal_seq2seq_gscan_amd.dataset import BatchStager, GroundedScanDataset
    from multimodal_seq2seq_gscan_amd.model import Model
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, write_dataset_file
    from multimodal_seq2seq_gscan_amd.train import TrainStep
    path = str(tmp_path / "dataset.txt")
    write_dataset_file(path, {"train": 150}, Shape(batch=1, max_command=8, max_target=12), seed=11)
    data = GroundedScanDataset(path, str(tmp_path), k=0, split="train", generate_vocabulary=True)
    data.read_dataset()
    cfg = model_kwargs("demo", input_vocabulary_size=data.input_vocabulary_size, num_cnn_channels=16,
                       target_vocabulary_size=data.target_vocabulary_size, cnn_dropout_p=0.0, encoder_dropout_p=0.0,
                       decoder_dropout_p=0.0)
    torch.manual_seed(3)
    staged_model, plain_model = Model(**cfg).cuda(), Model(**cfg).cuda()
    plain_model.load_state_dict(staged_model.state_dict())
    staged_step, plain_step = TrainStep(staged_model, learning_rate=1e-3), TrainStep(plain_model, learning_rate=1e-3)
    keys = ("commands", "cmd_lengths", "world", "targets", "tgt_lengths", "target_positions")
    stager = BatchStager(torch.device("cuda"), data.slab_bytes(16))
    np.random.seed(4)
    seen = 0
    for epoch in range(3):
        # the second epoch orders windows of batches by target length: batches narrower than the split's longest row
        data.shuffle_data(bucket_batches=3 if epoch == 1 else 0, batch_size=16)
        for b in data.batches(16, stager=stager):
            idx = b["index"]
            L, T = int(data._input_lengths[idx].max()), int(data._target_lengths[idx].max())
            assert b["world"].dtype == torch.uint8 and b["cmd_lengths"].dtype == torch.int32
            out = staged_step({k: b[k] for k in keys})                      # the step runs on the views ...
            host = {"commands": torch.from_numpy(data._commands[idx, :L]), "targets": torch.from_numpy(data._targets[idx, :T]),
                    "world": torch.from_numpy(data._grids[idx]).float(),
                    "cmd_lengths": torch.from_numpy(data._input_lengths[idx]), "tgt_lengths": torch.from_numpy(data._target_lengths[idx]),
                    "target_positions": torch.from_numpy(data._target_positions[idx])}
            ref = plain_step({k: v.cuda() for k, v in host.items()})         # ... and equals the float32 route
            assert torch.equal(b["commands"].cpu(), host["commands"]) and torch.equal(b["targets"].cpu(), host["targets"])
            assert torch.equal(b["world"].cpu(), torch.from_numpy(data._grids[idx]))
            assert torch.equal(b["cmd_lengths"].cpu().long(), host["cmd_lengths"])
            assert torch.equal(b["target_positions"].cpu(), host["target_positions"])
            assert abs(out["loss"].item() - ref["loss"].item()) < 1e-5
            seen += len(idx)
    assert seen == 3 * 150 and stager.count == 3 * 10
    # thirty Adam steps: the two models see bit-identical batches, but their split-K weight gradients are added with float
    # atomics (order varies run to run, ~1e-7) and Adam turns a sign flip of a near-zero gradient into a full
    # learning-rate step — so a handful of parameters may differ by a few 1e-4 (GSCAN_DETERMINISTIC=1: none does)
    diff = (staged_model.flat_parameters - plain_model.flat_parameters).abs()
    assert (diff > 1e-4).float().mean().item() < 2e-3 and diff.max().item() < 5e-3, (diff.max().item(), (diff > 1e-4).sum().item())
    data._order = np.arange(150)
    on_device = list(data.get_data_iterator(batch_size=32))
    on_host = list(data.get_data_iterator(batch_size=32, device=torch.device("cpu")))
    for d, h in zip(on_device, on_host):
        assert d[3].dtype == torch.float32 and torch.equal(d[3].cpu(), h[3]) and torch.equal(d[0].cpu(), h[0])
        assert torch.equal(d[5].cpu(), h[5]) and torch.equal(d[8].cpu(), h[8]) and (d[1] == h[1]).all()


def test_reference_checkpoint_scores_the_same_on_the_device(tmp_path):
    """§8 f4 on the device (model.py:228-235 then :206-219): `load_model` of the checkpoint the REFERENCE wrote, then
    `Model.forward` / `get_loss` on the HIP path, against the log-probabilities the reference itself computed after
    loading the same file (tests/golden/make_golden_checkpoint.py --logp-only)."""
    from multimodal_seq2seq_gscan_amd.model import Model
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fx = load_fixture("demo_reference_checkpoint_logp.npz")
    cfg = model_kwargs("demo", output_directory=str(tmp_path), cnn_hidden_num_channels=4, cnn_kernel_size=3)
    torch.manual_seed(123)                                    # different initial weights: everything must come from the file
    model = Model(**cfg).cuda()
    model.load_model(os.path.join(golden, "demo_reference_checkpoint.pth.tar"))
    assert model.trained_iterations == int(fx["iteration"]) == 2
    model.eval()
    batch = fixture_batch(fx)
    d = {k: v.cuda() for k, v in batch.items()}
    with torch.no_grad():
        logp, _ = model(commands_input=d["commands"], commands_lengths=batch["cmd_lengths"].tolist(),
                        situations_input=d["world"], target_batch=d["targets"],
                        target_lengths=batch["tgt_lengths"].tolist())
        loss = model.get_loss(logp, d["targets"])
    msg, err = report("logp after load_model", logp.cpu(), torch.from_numpy(fx["logp"]))
    assert err < TOL, msg
    assert abs(loss.item() - float(fx["loss"])) < TOL
    # and the other direction on the device: save from here, load into a fresh model, same scores bit for bit
    path = model.save_checkpoint("again.pth.tar", is_best=False, optimizer_state_dict={"state": {}, "param_groups": []})
    torch.manual_seed(7)
    again = Model(**cfg).cuda()
    again.load_model(path)
    again.eval()
    with torch.no_grad():
        logp2, _ = again(commands_input=d["commands"], commands_lengths=batch["cmd_lengths"].tolist(),
                         situations_input=d["world"], target_batch=d["targets"],
                         target_lengths=batch["tgt_lengths"].tolist())
    assert torch.equal(logp2, logp)


def test_batches_staged_per_rank_are_the_shards_of_the_global_batches(tmp_path):
    """Data parallelism stages only a rank's OWN rows of every global batch (`batches(row_shard=(rank, world))`): the
    rows must be exactly what train.shard_batch cuts from the full batch, the ranks together cover every batch once,
    and a trailing batch with fewer rows than ranks is dropped by every rank."""
    from multimodal_seq2seq_gscan_amd.dataset import GroundedScanDataset
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, write_dataset_file
    from multimodal_seq2seq_gscan_amd.train import shard_batch
    path = str(tmp_path / "dataset.txt")
    write_dataset_file(path, {"train": 66}, Shape(batch=1, max_command=8, max_target=12), seed=13)   # 4 x 16 + 2
    data = GroundedScanDataset(path, str(tmp_path), k=0, split="train", generate_vocabulary=True)
    data.read_dataset()
    np.random.seed(5)
    data.shuffle_data()
    world = 3
    full = [{k: v.clone() for k, v in b.items() if k != "index"} | {"index": b["index"].copy()} for b in data.batches(16)]
    assert [len(b["index"]) for b in full] == [16, 16, 16, 16, 2]
    per_rank = [[{k: (v.clone() if torch.is_tensor(v) else v.copy()) for k, v in b.items()}
                 for b in data.batches(16, row_shard=(r, world))] for r in range(world)]
    assert all(len(p) == 4 for p in per_rank)                       # the 2-row batch is dropped on every rank
    for i in range(4):
        rows = np.concatenate([per_rank[r][i]["index"] for r in range(world)])
        assert np.array_equal(rows, full[i]["index"])
        for r in range(world):
            lo, hi = r * 16 // world, (r + 1) * 16 // world
            mine = per_rank[r][i]
            want = shard_batch({k: v for k, v in full[i].items() if k != "index"}, r, world)
            L, T = mine["commands"].shape[1], mine["targets"].shape[1]          # a shard is padded to ITS longest rows
            assert mine["commands"].shape[0] == hi - lo
            assert torch.equal(mine["commands"], want["commands"][:, :L]) and (want["commands"][:, L:] == 0).all()
            assert torch.equal(mine["targets"], want["targets"][:, :T]) and (want["targets"][:, T:] == 0).all()
            assert torch.equal(mine["world"], want["world"]) and torch.equal(mine["cmd_lengths"], want["cmd_lengths"])
            assert torch.equal(mine["target_positions"], want["target_positions"])
