"""Size-independent properties of the HIP training step at the benchmark's full size (256 rows, 6x6x16 grid, hidden
100, T = 20; BASELINE.json configs[1]), where the CPU oracle is too slow to be the checker:

  * rows do not interact: permuting the batch permutes the log-probabilities bit for bit and leaves loss and gradients
    unchanged (up to the summation order of the loss and of the split-K atomics);
  * a row computed alone, or in a small batch, has the log-probabilities it has in the full batch;
  * padding is inert: extra PAD columns on the targets or the commands change neither the loss nor the live
    log-probabilities (the reference masks by length, seq2seq_model.py:62-88,129-137, and ignores PAD targets,
    model.py:100);
  * the backward pass is linear in the upstream gradient.

Run: pytest -m gpu."""
import pytest
import torch

from multimodal_seq2seq_gscan_amd.config import model_kwargs
from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch

pytestmark = pytest.mark.gpu
PAD = 0


@pytest.fixture(scope="module")
def setup():
    from multimodal_seq2seq_gscan_amd.model import Model
    torch.manual_seed(7)
    cfg = model_kwargs("compositional", auxiliary_task=True)
    model = Model(**cfg).cuda().eval()
    batch = make_batch(Shape(batch=256, ragged=True), seed=11)
    return model, batch


def flat_grad(model):
    return torch.cat([p.grad.detach().flatten() for _, p in model.named_parameters()]).cpu()


def step(model, batch, rows=None, weight=0.3):
    b = batch if rows is None else {k: v[rows] for k, v in batch.items()}
    d = {k: v.cuda() for k, v in b.items()}
    model.zero_grad()
    logp, aux = model(commands_input=d["commands"], commands_lengths=b["cmd_lengths"].tolist(),
                      situations_input=d["world"], target_batch=d["targets"], target_lengths=b["tgt_lengths"].tolist())
    loss = model.get_loss(logp, d["targets"]) + weight * model.get_auxiliary_loss(aux, d["target_positions"])
    loss.backward()
    torch.cuda.synchronize()
    return logp.detach().cpu(), aux.detach().cpu(), loss.item(), flat_grad(model)


def test_batch_permutation(setup):
    model, batch = setup
    logp, aux, loss, grad = step(model, batch)
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(3))
    logp_p, aux_p, loss_p, grad_p = step(model, batch, rows=perm)
    assert torch.equal(logp_p, logp[perm]), "a row's log-probabilities depend on its position in the batch"
    assert torch.equal(aux_p, aux[perm])
    assert abs(loss_p - loss) < 1e-5
    assert (grad_p - grad).abs().max().item() < 1e-5 + 1e-4 * grad.abs().max().item()


def test_rows_are_independent_of_the_batch_they_sit_in(setup):
    model, batch = setup
    logp, aux, _, _ = step(model, batch)
    for rows in (torch.tensor([0]), torch.tensor([5, 17, 200]), torch.arange(100, 133)):
        sub = {k: v[rows] for k, v in batch.items()}
        # the sub-batch keeps the full padded widths, so only the batch size changes
        logp_s, aux_s, _, _ = step(model, sub)
        assert (logp_s - logp[rows]).abs().max().item() < 2e-6
        assert (aux_s - aux[rows]).abs().max().item() < 2e-6


def test_padding_is_inert(setup):
    model, batch = setup
    logp, aux, loss, grad = step(model, batch, weight=0.0)
    T, L = batch["targets"].shape[1], batch["commands"].shape[1]
    wide = dict(batch)
    wide["targets"] = torch.cat([batch["targets"], torch.full((256, 4), PAD, dtype=torch.long)], dim=1)
    wide["commands"] = torch.cat([batch["commands"], torch.full((256, 2), PAD, dtype=torch.long)], dim=1)
    logp_w, aux_w, loss_w, grad_w = step(model, wide, weight=0.0)
    assert logp_w.shape[1] == T + 4 and wide["commands"].shape[1] == L + 2
    # positions whose target is a live token: identical up to the changed GEMM tiling of the wider products
    live = torch.arange(T).unsqueeze(0) < (batch["tgt_lengths"] - 1).unsqueeze(1)
    assert (logp_w[:, :T][live] - logp[live]).abs().max().item() < 1e-5
    assert abs(loss_w - loss) < 1e-5
    assert (grad_w - grad).abs().max().item() < 1e-5 + 1e-4 * grad.abs().max().item()
    # the auxiliary head sums the visual attention over ALL decoder steps, padded ones included (seq2seq_model.py:479,
    # model.py:205): wider targets change it in the reference too, so the auxiliary term is left out here (weight 0)


def test_backward_is_linear_in_the_upstream_gradient(setup):
    model, batch = setup
    d = {k: v.cuda() for k, v in batch.items()}

    def grad_for(scale_logp, scale_aux):
        model.zero_grad()
        logp, aux = model(commands_input=d["commands"], commands_lengths=batch["cmd_lengths"].tolist(),
                          situations_input=d["world"], target_batch=d["targets"],
                          target_lengths=batch["tgt_lengths"].tolist())
        gen = torch.Generator(device="cuda").manual_seed(5)
        up_logp = torch.randn(logp.shape, device="cuda", generator=gen) * 1e-3
        up_aux = torch.randn(aux.shape, device="cuda", generator=gen) * 1e-3
        torch.autograd.backward([logp, aux], [scale_logp * up_logp, scale_aux * up_aux])
        torch.cuda.synchronize()
        return flat_grad(model)

    g10, g01, g23 = grad_for(1.0, 0.0), grad_for(0.0, 1.0), grad_for(2.0, 3.0)
    ref = 2.0 * g10 + 3.0 * g01
    assert (g23 - ref).abs().max().item() < 1e-6 + 1e-4 * ref.abs().max().item()


def test_dropout_drawn_inside_the_kernels_equals_the_same_masks_in_memory():
    """SURVEY.md 7 hard part 3 (production mode: a counter RNG inside the kernels).  A training-mode step draws its dropout
    where it is applied (csrc/dropout.h) and no mask exists in memory; gscan_dropout_masks_kernel_layout writes the masks
    of the same (seed, Philox stream) to memory, and the step fed with them through the pointer form is the same step: the
    forward pass bit for bit, the gradients up to the order of the float atomics.  Odd row counts (B*L and B*T are not
    multiples of four: partial row quads) and a ragged batch."""
    from multimodal_seq2seq_gscan_amd.model import Model, _dropout_in_kernel
    assert _dropout_in_kernel(), "GSCAN_DROPOUT_IN_KERNEL=0 in the environment of the test run"
    torch.manual_seed(5)
    for shape in (Shape(batch=37, ragged=True, max_target=19), Shape(batch=256)):
        model = Model(**model_kwargs("compositional", auxiliary_task=True)).cuda().train()
        batch = make_batch(shape, seed=13)
        B, L = batch["commands"].shape
        T, M = batch["targets"].shape[1], batch["world"].shape[1] ** 2
        calls = model._dropout_calls
        masks = model._draw_masks(B, L, T, M, torch.device("cuda"), materialize=True)
        model._dropout_calls = calls                      # rewind: the step below draws the same masks in its kernels
        for m, p in zip(masks, model.dropout_p):
            keep = 1.0 / (1.0 - p)
            assert bool(((m == 0) | ((m - keep).abs() < 1e-6)).all())
            assert abs((m == 0).float().mean().item() - p) < 0.02
        assert model._mask_buffer is None                 # nothing was drawn into memory by the model itself
        logp_k, aux_k, loss_k, grad_k = step(model, batch)
        model.set_dropout_masks(*masks)
        logp_m, aux_m, loss_m, grad_m = step(model, batch)
        assert torch.equal(logp_k, logp_m), "in-kernel dropout and the same masks in memory give different forward passes"
        assert abs(loss_k - loss_m) < 1e-6
        assert (grad_k - grad_m).abs().max().item() < 1e-6 + 1e-5 * grad_m.abs().max().item()
        logp_2, _, _, _ = step(model, batch)
        assert not torch.equal(logp_2, logp_k), "the next step drew the same dropout again"
        model.eval()
        logp_e, _, _, _ = step(model, batch)
        assert not torch.equal(logp_e, logp_k)            # and eval mode draws none


def test_training_memorises_one_batch():
    """End-to-end sanity of gradients + fused Adam/LR on the device: with dropout off, a few hundred steps on one
    fixed batch of random targets drive the loss from ~log(V) to near zero and the exact-match rate to 100 %."""
    from multimodal_seq2seq_gscan_amd.model import Model
    from multimodal_seq2seq_gscan_amd.train import TrainStep
    torch.manual_seed(3)
    cfg = model_kwargs("compositional", cnn_dropout_p=0.0, encoder_dropout_p=0.0, decoder_dropout_p=0.0)
    model = Model(**cfg).cuda()
    batch = {k: v.cuda() for k, v in make_batch(Shape(batch=16, max_target=12, ragged=True), seed=5).items()}
    step = TrainStep(model, learning_rate=3e-3, lr_decay_steps=100000.0)
    first = float(step(batch)["loss"].item())
    for _ in range(500):
        out = step(batch)
    last = float(out["loss"].item())
    assert first > 1.0 and last < 0.05, (first, last)
    model.eval()
    logp, _ = model(commands_input=batch["commands"], commands_lengths=batch["cmd_lengths"].tolist(),
                    situations_input=batch["world"], target_batch=batch["targets"],
                    target_lengths=batch["tgt_lengths"].tolist())
    accuracy, exact_match = model.get_metrics(logp, batch["targets"])
    assert exact_match == 100.0 and accuracy > 99.9, (accuracy, exact_match)


def _forward_logp(model, batch):
    model.eval()
    with torch.no_grad():
        logp, _ = model(commands_input=batch["commands"], commands_lengths=batch["cmd_lengths"], situations_input=batch["world"],
                        target_batch=batch["targets"], target_lengths=batch["tgt_lengths"])
    return logp


def _selfserved(model, batch):
    """Chunks of the convolution weight image that waiting world-encoder workgroups wrote themselves (conv.hip): the
    word behind the 512 flags of the fused prologue launch in the workspace, a running count."""
    B, L = batch["commands"].shape
    dims = model._dims(B, L, batch["targets"].shape[1], batch["world"].shape[1])
    return int(model.workspace_view(dims, "conv_flags").view(torch.int32)[512].item())


def test_fused_prologue_launch_on_eight_compute_units(tmp_path):
    """The prologue + world encoder launch (conv.hip) waits, inside the launch, for the workgroups that write the
    convolution weight image.  On a stream restricted to EIGHT compute units (hipExtStreamCreateWithCUMask) a few
    thousand workgroups queue for 8 CUs: 1 000 forward passes must finish and equal the unrestricted result bit for
    bit — the wait is bounded and cannot depend on dispatch order (a waiter that runs out of patience writes the chunk
    itself)."""
    import ctypes as C
    from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
    from multimodal_seq2seq_gscan_amd.model import Model
    cfg = model_kwargs("compositional")
    torch.manual_seed(3)
    model = Model(**cfg).cuda()
    shape = Shape(batch=64, grid=6, channels=16, input_vocab=cfg["input_vocabulary_size"],
                  target_vocab=cfg["target_vocabulary_size"], max_command=10, max_target=12)
    batch = {k: v.cuda() for k, v in make_batch(shape, 11).items()}
    ref = _forward_logp(model, batch).clone()
    hip = C.CDLL("libamdhip64.so")
    handle = C.c_void_p()
    mask = (C.c_uint32 * 8)(0xFF, 0, 0, 0, 0, 0, 0, 0)            # CUs 0..7 of 256
    assert hip.hipExtStreamCreateWithCUMask(C.byref(handle), 8, mask) == 0
    stream = torch.cuda.ExternalStream(handle.value)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        for i in range(1000):
            out = _forward_logp(model, batch)
            if i % 250 == 0:
                assert torch.equal(out, ref), i
        assert torch.equal(out, ref)
    torch.cuda.synchronize()
    assert hip.hipStreamDestroy(handle) == 0


def test_fused_prologue_waiters_serve_themselves_when_the_image_workgroups_do_nothing(tmp_path):
    """GSCAN_FUSED_SKIP_IMAGE=1 (test hook) makes the image workgroups of the fused launch return at once — what an
    adversarial dispatch order would look like to the world encoder's workgroups.  After a short wait
    (GSCAN_FUSED_LATE_AFTER polls) they write the missing chunks themselves: same log-probabilities, bit for bit, and
    the self-service counter moves."""
    import os, subprocess, sys
    worker = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
from test_properties_gpu import _forward_logp, _selfserved, model_kwargs
from multimodal_seq2seq_gscan_amd.model import Model
from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
cfg = model_kwargs("compositional")
torch.manual_seed(3)
model = Model(**cfg).cuda()
shape = Shape(batch=32, grid=6, channels=16, input_vocab=cfg["input_vocabulary_size"], target_vocab=cfg["target_vocabulary_size"], max_command=10, max_target=12)
batch = {k: v.cuda() for k, v in make_batch(shape, 11).items()}
out = _forward_logp(model, batch)
before = _selfserved(model, batch)
for _ in range(20):
    out = _forward_logp(model, batch)
torch.save({"logp": out.cpu(), "selfserved": _selfserved(model, batch) - before}, sys.argv[3])
"""
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for name, extra in (("normal", {}), ("skip", {"GSCAN_FUSED_SKIP_IMAGE": "1", "GSCAN_FUSED_LATE_AFTER": "50"})):
        path = str(tmp_path / f"{name}.pt")
        r = subprocess.run([sys.executable, "-c", worker, os.path.dirname(here), here, path], env=dict(os.environ, **extra),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        res[name] = torch.load(path)
    assert torch.equal(res["normal"]["logp"], res["skip"]["logp"])
    assert res["normal"]["selfserved"] == 0, res["normal"]["selfserved"]
    assert res["skip"]["selfserved"] > 0


@pytest.mark.parametrize("variant", ["one_encoder_layer", "two_encoder_layers", "sums_over_time_first"])
def test_deterministic_mode_gives_bitwise_reproducible_training_steps(tmp_path, variant):
    """GSCAN_DETERMINISTIC=1: every sum the step forms across workgroups is added in a fixed order (split-K partial tiles
    through slabs, embedding gradients by one workgroup per vocabulary chunk), so the same step from the same state gives
    the same bits: loss, every gradient, and the parameters after three Adam steps.  (The default mode adds with float
    atomics: results differ in the last bits from run to run.)"""
    import os, subprocess, sys
    worker = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
from multimodal_seq2seq_gscan_amd.config import model_kwargs
from multimodal_seq2seq_gscan_amd.model import Model
from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
from multimodal_seq2seq_gscan_amd.train import TrainStep
cfg = model_kwargs("compositional", auxiliary_task=True)
if os.environ.get("GSCAN_TEST_DEEP_ENCODER") == "1":
    # two bidirectional encoder layers: both directions' dX products of the upper layer add into ONE buffer from one launch
    # (step.hip, split_k < 0) - one slice each in this mode, two commutative adds per element (ADVICE r4)
    cfg["num_encoder_layers"] = 2
shape = Shape(batch=96, input_vocab=cfg["input_vocabulary_size"], target_vocab=cfg["target_vocabulary_size"], ragged=True)
batch = {k: v.cuda() for k, v in make_batch(shape, 5).items() if k in ("commands", "cmd_lengths", "world", "targets", "target_positions")}
runs = []
for rep in range(3):
    torch.manual_seed(7)
    model = Model(**cfg).cuda()
    step = TrainStep(model, learning_rate=1e-3, weight_target_loss=0.3)
    grads = []
    step.on_gradients = lambda g: grads.append(g.detach().clone())
    losses = [step(batch)["loss"].item() for _ in range(3)]
    torch.cuda.synchronize()
    runs.append({"losses": losses, "grads": [g.cpu() for g in grads], "params": model.flat_parameters.detach().cpu().clone()})
    step.close()
torch.save(runs, sys.argv[3])
"""
    here = os.path.dirname(os.path.abspath(__file__))
    path = str(tmp_path / "runs.pt")
    r = subprocess.run([sys.executable, "-c", worker, os.path.dirname(here), here, path],
                       # sums_over_time_first: the long-target backward pass (csrc/step.hip attention_time_reduced) forced on
                       env=dict(os.environ, GSCAN_DETERMINISTIC="1", GSCAN_TEST_DEEP_ENCODER="1" if variant == "two_encoder_layers" else "0",
                                **({"GSCAN_TIME_REDUCED_T": "1"} if variant == "sums_over_time_first" else {})),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    runs = torch.load(path)
    for other in runs[1:]:
        assert other["losses"] == runs[0]["losses"]
        for g0, g1 in zip(runs[0]["grads"], other["grads"]):
            assert torch.equal(g0, g1), (g0 - g1).abs().max().item()
        assert torch.equal(other["params"], runs[0]["params"])


def test_file_fed_training_loop_keeps_up_with_resident_batches(tmp_path):
    """SURVEY.md 8 f1 / VERDICT r4 item 3: the training step fed by the dataset reader and the staged batcher runs at the
    speed of the same step on batches resident in HBM — 97 % at the benchmark shape (bench.py --with-batcher) — and a
    fall below 93 % fails here.  Paper dims, 256-row batches of a generated dataset file in the reference's order and in
    length buckets (every batch another padded shape: the dropout masks drawn ahead by the optimiser launch must serve
    any shape, Model._draw_masks; that order must be FASTER per batch than the reference order).  A timing test: it runs
    in a process of its own (inside the suite's process, hundreds of tests in, the host side of the loop runs at 2/3 of
    its speed), whole passes over the file after a warm-up pass, the best of up to four passes counts."""
    import os, subprocess, sys
    worker = r"""
import sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from multimodal_seq2seq_gscan_amd.config import model_kwargs
from multimodal_seq2seq_gscan_amd.dataset import BatchStager, GroundedScanDataset
from multimodal_seq2seq_gscan_amd.model import Model
from multimodal_seq2seq_gscan_amd.synthetic import Shape, write_dataset_file
from multimodal_seq2seq_gscan_amd.train import TrainStep
tmp = sys.argv[2]
B = 256
path = tmp + "/dataset.txt"
write_dataset_file(path, {"train": 30000}, Shape(batch=1, max_command=10, max_target=20), seed=7)
data = GroundedScanDataset(path, tmp, k=0, split="train", generate_vocabulary=True)
data.read_dataset()
cfg = model_kwargs("compositional", input_vocabulary_size=data.input_vocabulary_size,
                   target_vocabulary_size=data.target_vocabulary_size, num_cnn_channels=data.image_channels)
torch.manual_seed(1)
model = Model(**cfg).cuda()
step = TrainStep(model, learning_rate=1e-3)
stager = BatchStager(torch.device("cuda"), data.slab_bytes(B))
keys = ("commands", "cmd_lengths", "world", "targets", "tgt_lengths", "target_positions")
np.random.seed(2)

def fed_epoch(bucket):
    data.shuffle_data(bucket_batches=bucket, batch_size=B)
    n, t0, last, shapes = 0, None, None, set()
    for i, b in enumerate(data.batches(B, stager=stager)):
        if b["commands"].shape[0] != B:
            continue
        if i == 30:
            torch.cuda.synchronize()
            t0, n = time.perf_counter(), 0
        step({k: b[k] for k in keys})
        n, last = n + 1, b
        shapes.add((b["commands"].shape[1], b["targets"].shape[1]))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, {k: last[k].clone() for k in keys}, shapes, n

def resident_loop(batch, n):
    for _ in range(20):
        step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step(batch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

fed_epoch(0)                                             # allocator blocks, slab views, kernel attributes
ratios, fed_ref = [], []
for attempt in range(4):
    fed, last, _, n = fed_epoch(0)
    ratios.append(resident_loop(last, n) / fed)
    fed_ref.append(fed)
    if ratios[-1] >= 0.95:
        break
fed_epoch(8)
fed_b, _, shapes, _ = min((fed_epoch(8) for _ in range(2)), key=lambda r: r[0])
step.close()
print("RESULT", max(ratios), min(fed_ref), fed_b, len(shapes), flush=True)
"""
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-c", worker, os.path.dirname(here), str(tmp_path)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "RESULT" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    ratio, fed_ref, fed_bucketed, shapes = [float(x) for x in r.stdout.split("RESULT")[1].split()[:4]]
    assert ratio >= 0.93, f"file-fed loop at {100 * ratio:.1f} % of resident batches"
    # length buckets: fewer padded decoder steps per batch, every batch another shape — faster per batch than the
    # reference order (the resident comparison batch would be ONE of those shapes: not a yardstick here)
    assert shapes > 3
    assert fed_bucketed < fed_ref, (fed_bucketed, fed_ref)
