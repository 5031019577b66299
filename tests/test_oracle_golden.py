"""Pin the CPU oracle against outputs of the reference itself (tests/golden/*.npz).

Tolerance: the north star asks for 1e-4 on loss/log-probabilities in fp32; the oracle is
held to 2e-5 so that it is a tighter yardstick than the thing it measures."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, fixture_batch, fixture_params, load_fixture, rel_err
from multimodal_seq2seq_gscan_amd.config import PARAMETER_TOTALS, model_kwargs
from oracle import seq2seq_oracle as oracle
from weights import parameter_shapes

TOL = 2e-5


def _check_case(name, cfg, grads="full"):
    fx = load_fixture(name)
    p = fixture_params(cfg, fx)
    batch = fixture_batch(fx)
    loss, g, logp = oracle.loss_and_grads(p, batch, conditional=cfg["conditional_attention"],
                                          auxiliary=cfg["auxiliary_task"],
                                          bidirectional=cfg["encoder_bidirectional"],
                                          weight_target_loss=float(fx["weight_target_loss"]))
    assert torch.allclose(logp, torch.from_numpy(fx["logp"]), atol=TOL, rtol=0)
    assert abs(loss.item() - float(fx["loss"])) < TOL
    acc, exact = oracle.metrics(logp, batch["targets"])
    assert abs(acc - float(fx["accuracy"])) < 1e-3 and abs(exact - float(fx["exact_match"])) < 1e-3
    for k, v in g.items():
        if "grad/" + k in fx:
            ref = torch.from_numpy(fx["grad/" + k])
            assert torch.allclose(v, ref, atol=TOL, rtol=1e-4), k
        if "gradnorm/" + k in fx:
            assert abs(v.double().norm().item() - float(fx["gradnorm/" + k])) < 1e-4 * max(1.0, float(fx["gradnorm/" + k])), k


@pytest.mark.parametrize("cond", [True, False])
@pytest.mark.parametrize("aux", [True, False])
def test_demo_variants(cond, aux):
    cfg = model_kwargs("demo", conditional_attention=cond, auxiliary_task=aux)
    _check_case(f"demo_cond{int(cond)}_aux{int(aux)}.npz", cfg)


def test_compositional_all_grads():
    _check_case("compositional_b16.npz", model_kwargs("compositional"))


def test_geca_aux():
    _check_case("geca_aux_b16.npz", model_kwargs("compositional", auxiliary_task=True))


def test_target_length_t120():
    _check_case("target_length_t120.npz", model_kwargs("target_length"))


DEEP_ENCODERS = {
    "demo_enc2.npz": dict(num_encoder_layers=2, auxiliary_task=True),
    "demo_enc3_unidirectional.npz": dict(num_encoder_layers=3, conditional_attention=False,
                                         encoder_bidirectional=False),
}


@pytest.mark.parametrize("name", sorted(DEEP_ENCODERS))
def test_more_than_one_encoder_layer(name):
    """nn.LSTM(num_layers=n): layer inputs are the concatenated directions of the layer below, the direction sums
    and the final state come from the last layer (seq2seq_model.py:44-45,76-82)."""
    _check_case(name, model_kwargs("demo", **DEEP_ENCODERS[name]))


def test_dropout_host_masks():
    """The reference draws its dropout masks CNN -> encoder embedding -> decoder embedding per step
    (length-sorted rows); with those masks handed over, the oracle reproduces the train-mode output."""
    cfg = model_kwargs("demo")
    fx = load_fixture("demo_dropout_hostmask.npz")
    p = fixture_params(cfg, fx)
    batch = fixture_batch(fx)
    masks = tuple(torch.from_numpy(fx[k]) for k in ("mask_cnn", "mask_enc", "mask_dec"))
    logp, _ = oracle.forward(p, batch["commands"], batch["cmd_lengths"], batch["world"], batch["targets"],
                             masks=masks)
    assert torch.allclose(logp, torch.from_numpy(fx["logp"]), atol=TOL, rtol=0)
    assert abs(oracle.sequence_loss(logp, batch["targets"]).item() - float(fx["loss"])) < TOL


def test_adam_three_steps():
    """Loop body of seq2seq/train.py:96-114 for three iterations: Adam + LambdaLR + update_state."""
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
    cfg = model_kwargs("demo")
    fx = load_fixture("demo_adam3.npz")
    p = fixture_params(cfg, {"seed_weights": 11})
    names = list(p.keys())
    m = [torch.zeros_like(p[k]) for k in names]
    v = [torch.zeros_like(p[k]) for k in names]
    shape = Shape(batch=4, grid=4, channels=15, input_vocab=14, target_vocab=6, max_command=7, max_target=10,
                  ragged=True)
    for step in range(3):
        batch = make_batch(shape, 100 + step)
        loss, g, _ = oracle.loss_and_grads(p, batch)
        assert abs(loss.item() - float(fx["losses"][step])) < TOL
        oracle.adam_step([p[k] for k in names], [g[k] for k in names], m, v, step + 1, float(fx["lr"]),
                         lr_decay=float(fx["lr_decay"]), lr_decay_steps=float(fx["lr_decay_steps"]))
    # Adam divides by sqrt(v)+1e-8: an element whose gradient is rounding noise (|g| ~ 1e-9) moves by a
    # fraction of lr that depends on that noise, hence 1e-5 (= lr/100) rather than 1e-6 here.
    for k in names:
        assert torch.allclose(p[k], torch.from_numpy(fx["param/" + k]), atol=1e-5, rtol=0), k
    assert int(fx["trained_iterations"]) == 3


def test_known_parameter_totals():
    """Known answers the reference publishes: README.md:264, adverb_run_1.txt:58, target_lengths_run_1.txt:79."""
    with open(os.path.join(GOLDEN, "init_seed42.json")) as f:
        init = json.load(f)
    for workload, total in PARAMETER_TOTALS.items():
        shapes = parameter_shapes(model_kwargs(workload))
        assert sum(int(np.prod(s)) for s in shapes.values()) == total
        assert init[workload]["total"] == total
        assert list(init[workload]["params"].keys()) == list(shapes.keys())
        for k, s in shapes.items():
            assert tuple(init[workload]["params"][k]["shape"]) == tuple(s)


def test_oracle_greedy_decode_matches_reference_predict_loop():
    """oracle.greedy_decode against the reference's own encode_input / decode_input loop (predict.py:82-115, one
    example at a time): same tokens, same stopping step, logits and attention rows within tolerance."""
    fx = load_fixture("demo_greedy.npz")
    cfg = model_kwargs("demo", conditional_attention=True, auxiliary_task=True)
    params = fixture_params(cfg, fx)
    batch = fixture_batch(fx)
    rows = oracle.greedy_decode(params, batch["commands"], batch["cmd_lengths"], batch["world"], int(fx["sos"]),
                                int(fx["eos"]), int(fx["max_steps"]), conditional=True)
    for r, row in enumerate(rows):
        n = int(fx["nsteps"][r])
        assert row["tokens"] == fx["tokens"][r, :n].tolist(), r
        L = int(batch["cmd_lengths"][r])
        assert torch.allclose(torch.stack(row["logits"]), torch.from_numpy(fx["logits"][r, :n]), atol=2e-5)
        assert torch.allclose(torch.stack(row["alpha_text"])[:, :L], torch.from_numpy(fx["alpha_text"][r, :n, :L]), atol=2e-5)
        assert torch.allclose(torch.stack(row["alpha_vis"]), torch.from_numpy(fx["alpha_vis"][r, :n]), atol=2e-5)


def test_oracle_scores_the_reference_checkpoint_as_the_reference_does():
    """The weights of the checkpoint the reference WROTE, scored by the oracle, against the log-probabilities the
    reference computed after loading that file itself (make_golden_checkpoint.py --logp-only)."""
    fx = load_fixture("demo_reference_checkpoint_logp.npz")
    ck = torch.load(os.path.join(GOLDEN, "demo_reference_checkpoint.pth.tar"), map_location="cpu", weights_only=False)
    loss, _, logp = oracle.loss_and_grads(dict(ck["state_dict"]), fixture_batch(fx), conditional=True, auxiliary=False)
    assert abs(loss.item() - float(fx["loss"])) < TOL
    assert (logp - torch.from_numpy(fx["logp"])).abs().max().item() < TOL
