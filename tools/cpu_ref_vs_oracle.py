"""time the IMPORTED REFERENCE beside this repository's CPU oracle — build container only.

    python tools/cpu_ref_vs_oracle.py [--threads 8] [--steps 5] [--json profiles/cpu_ref_vs_oracle_latest.json] > profiles/rNN_cpu_ref_vs_oracle.txt

--json writes the anchor bench.py puts on its line as cpu_baseline.reference_over_port (ratios, date, threads and a hash of
the oracle's source: a later edit of the oracle shows up there as oracle_unchanged_since = false).

The reference (`/root/reference/seq2seq/model.py`, read-only, never copied) is imported, seeded with the same golden
weights as the oracle, and both run the same training step — forward, loss, backward, Adam + LR step, dropout at the
paper values — on the same synthetic S1 / S3 batches.  Printed: loss agreement of the first step (dropout off) and
examples/s of both.  This is what anchors `bench.py`'s `cpu_baseline` (kind "port": the oracle, the only one of the two
that can travel to the GPU box) to the reference's own CPU path.

The script refuses to run where /root/reference does not exist; nothing of the reference is written anywhere."""
from __future__ import annotations

import argparse
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
if not os.path.isdir(os.path.join(REFERENCE, "seq2seq")):
    raise SystemExit(f"{REFERENCE} is not here: this comparison runs in the build container only "
                     f"(the reference never travels; bench.py times the oracle on the GPU box)")
sys.dont_write_bytecode = True
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
warnings.filterwarnings("ignore")

import torch  # noqa: E402

from multimodal_seq2seq_gscan_amd.config import model_kwargs  # noqa: E402
from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch  # noqa: E402
from oracle import seq2seq_oracle as oracle  # noqa: E402   (checker only)
from weights import golden_weights  # noqa: E402


def import_reference():
    sys.path.insert(0, REFERENCE)
    for name in [m for m in sys.modules if m == "seq2seq" or m.startswith("seq2seq.")]:
        del sys.modules[name]                 # this repository ships a drop-in package of the same name
    from seq2seq.model import Model
    assert Model.__module__ == "seq2seq.model" and "/root/reference" in sys.modules["seq2seq.model"].__file__
    return Model


def reference_step(ReferenceModel, cfg, weights, batch, steps, train_mode):
    model = ReferenceModel(**cfg)
    model.load_state_dict(weights, strict=False)
    model.train(train_mode)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3, betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda t: 0.9 ** (t / 20000.0))

    def one():
        logp, _ = model(commands_input=batch["commands"], commands_lengths=batch["cmd_lengths"].tolist(),
                        situations_input=batch["world"], target_batch=batch["targets"],
                        target_lengths=batch["tgt_lengths"].tolist())
        loss = model.get_loss(logp, batch["targets"])
        loss.backward()
        opt.step(); sched.step(); opt.zero_grad()
        model.update_state(is_best=False)
        return float(loss)
    first = one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    return first, steps * batch["commands"].shape[0] / (time.perf_counter() - t0)


def oracle_step(cfg, weights, batch, steps, train_mode):
    params = {k: v.clone() for k, v in weights.items()}
    names = list(params)
    m = [torch.zeros_like(params[k]) for k in names]
    v = [torch.zeros_like(params[k]) for k in names]
    B, L = batch["commands"].shape
    T, G = batch["targets"].shape[1], batch["world"].shape[1]
    p = (cfg["cnn_dropout_p"], cfg["encoder_dropout_p"], cfg["decoder_dropout_p"])
    drop = torch.nn.functional.dropout

    def one(step):
        masks = None
        if train_mode:
            masks = (drop(torch.ones(B, G * G, 3 * cfg["cnn_hidden_num_channels"]), p[0]),
                     drop(torch.ones(B, L, cfg["embedding_dimension"]), p[1]),
                     drop(torch.ones(B, T, cfg["decoder_hidden_size"]), p[2]))
        loss, g, _ = oracle.loss_and_grads(params, batch, conditional=cfg["conditional_attention"],
                                           auxiliary=False, masks=masks)
        oracle.adam_step([params[k] for k in names], [g[k] for k in names], m, v, step, 1e-3)
        return float(loss)
    first = one(1)
    t0 = time.perf_counter()
    for i in range(steps):
        one(2 + i)
    return first, steps * B / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    anchor = {}
    torch.set_num_threads(args.threads)
    ReferenceModel = import_reference()
    print(f"# reference (imported from {REFERENCE}) vs oracle/seq2seq_oracle.py, torch {torch.__version__} CPU, "
          f"{args.threads} threads, {args.steps} timed steps after one warm-up step; step = forward + loss + backward + Adam + LR")
    print(f"{'workload':34} {'loss ref':>10} {'loss oracle':>12} {'|diff|':>9} {'ref ex/s':>9} {'oracle ex/s':>12} {'oracle/ref':>10}")
    for label, workload, shape_kw in (("S1 (B=256, k=7, T=20)", "compositional", dict(batch=256)),
                                      ("S3 (B=256, k=13, T=120)", "target_length", dict(batch=256, max_target=120))):
        cfg = model_kwargs(workload)
        shape = Shape(grid=6, channels=cfg["num_cnn_channels"], input_vocab=cfg["input_vocabulary_size"],
                      target_vocab=cfg["target_vocabulary_size"], **shape_kw)
        batch = make_batch(shape, 1234)
        weights = {k: torch.from_numpy(w) for k, w in golden_weights(cfg, 1).items()}
        # agreement with dropout off (the two draw their masks differently), speed at the paper's dropout
        l_ref, _ = reference_step(ReferenceModel, cfg, weights, batch, 0, False)
        l_orc, _ = oracle_step(cfg, weights, batch, 0, False)
        _, s_ref = reference_step(ReferenceModel, cfg, weights, batch, args.steps, True)
        _, s_orc = oracle_step(cfg, weights, batch, args.steps, True)
        print(f"{label:34} {l_ref:10.6f} {l_orc:12.6f} {abs(l_ref - l_orc):9.1e} {s_ref:9.1f} {s_orc:12.1f} {s_orc / s_ref:10.2f}")
        anchor[label.split()[0]] = {"reference_examples_per_s": round(s_ref, 1), "port_examples_per_s": round(s_orc, 1),
                                    "reference_over_port": round(s_ref / s_orc, 3), "loss_abs_diff": abs(l_ref - l_orc)}
    if args.json:
        import datetime, hashlib, json
        with open(os.path.join(ROOT, "oracle", "seq2seq_oracle.py"), "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()[:16]
        anchor.update(measured="build container (8 CPUs), imported /root/reference beside oracle/seq2seq_oracle.py, "
                               f"{args.threads} threads, {args.steps} timed steps, torch {torch.__version__}",
                      date=datetime.date.today().isoformat(), oracle_sha=sha)
        with open(args.json, "w") as f:
            json.dump(anchor, f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main()
