"""Randomised parity sweep (GPU): random model configurations and batch shapes, HIP step against the CPU oracle
(log-probabilities, loss, every gradient; tolerance of tests/test_parity_gpu.py).  Not part of the test-suite: run it
after changes to a kernel's indexing.
    python tools/fuzz_parity.py [--cases 40] [--seed 0]"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch

from multimodal_seq2seq_gscan_amd.config import model_kwargs
from multimodal_seq2seq_gscan_amd.model import Model
from multimodal_seq2seq_gscan_amd.synthetic import Shape, make_batch
from oracle import seq2seq_oracle as oracle
from weights import golden_weights

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=40)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--extremes", action="store_true", help="a fixed list of degenerate / limit shapes instead of random ones")
ap.add_argument("--wide", action="store_true",
                help="sample the whole range the reference accepts (hidden sizes to 256 incl. sizes that are not multiples "
                     "of 4, grids to 12x12, commands to 128 tokens): most cases then run on the streaming kernels")
ap.add_argument("--train-step", action="store_true",
                help="also run train.TrainStep's ONE library call (gscan_train_step_nll, learning rate 0) on a second model "
                     "with the same weights: its loss and gradients must equal the oracle's too")
args = ap.parse_args()
# (H, He, E, k, Co, cond, aux, bi, layers, B, G, L, T)
EXTREMES = [
    (100, 100, 25, 7, 50, 1, 0, 1, 1, 1, 2, 1, 1),       # 2x2 cells (the generator needs an agent and an object), one token, one step
    (100, 100, 25, 7, 50, 1, 1, 1, 1, 2, 2, 2, 1),
    (100, 100, 25, 7, 50, 1, 0, 1, 1, 1, 6, 64, 2),      # the longest command supported
    (100, 100, 25, 7, 50, 1, 0, 1, 1, 2, 8, 10, 3),      # the largest grid supported (gate images streamed from L2)
    (100, 100, 25, 7, 50, 1, 0, 1, 1, 1, 8, 64, 2),      # both at once: 240 KB of LDS per row in the resident kernels -> streaming kernels
    (256, 256, 64, 7, 50, 1, 1, 1, 2, 2, 12, 128, 3),    # round 4: the widest of everything at once (streaming kernels)
    (255, 255, 63, 7, 50, 1, 0, 1, 1, 2, 3, 5, 3),       # odd widths
    (1, 1, 1, 1, 1, 1, 1, 0, 1, 2, 2, 2, 2),             # hidden size 1
    (100, 100, 25, 13, 50, 0, 0, 0, 1, 3, 2, 1, 40),
    (4, 4, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2),             # every width at its minimum
    (100, 128, 64, 7, 50, 1, 0, 1, 2, 2, 6, 10, 5),      # embedding wider than one 32-column block of the projection
    (64, 64, 33, 3, 70, 1, 1, 1, 3, 5, 7, 9, 3),
    (100, 100, 25, 7, 200, 1, 0, 1, 1, 2, 6, 10, 20),    # 200 output channels: four 64-lane chunks in the world encoder
    (96, 100, 25, 7, 50, 1, 0, 1, 1, 257, 6, 10, 4),     # one row more than the chip has CUs
    (1024, 64, 300, 3, 8, 1, 1, 1, 1, 2, 3, 4, 2),       # round 5: the widest decoder / embedding the library takes
    (700, 300, 513, 3, 8, 0, 0, 0, 2, 1, 2, 3, 2),       # embedding gradients in three column blocks
]
if args.extremes:
    args.cases = len(EXTREMES)
rng = random.Random(args.seed)
TOL = 1e-4
bad = 0
for case in range(args.cases):
    H = rng.choice(list(range(4, 101, 4)))              # every compiled decoder size
    He = rng.choice(list(range(4, 129, 4)))             # every compiled encoder size
    if args.wide:
        H = rng.choice([rng.randint(1, 256), rng.choice([128, 200, 256]), rng.choice(list(range(4, 101, 4)))])
        He = rng.choice([rng.randint(1, 256), rng.choice([128, 200, 256]), rng.choice(list(range(4, 129, 4)))])
    if args.extremes:
        xH, xHe, xE, xk, xCo, xcond, xaux, xbi, xlayers, xB, xG, xL, xT = EXTREMES[case]
        H, He = xH, xHe
        cfg = model_kwargs("demo", cnn_dropout_p=0.0, encoder_dropout_p=0.0, decoder_dropout_p=0.0,
                           decoder_hidden_size=xH, encoder_hidden_size=xHe, embedding_dimension=xE, cnn_kernel_size=xk,
                           cnn_hidden_num_channels=xCo, conditional_attention=bool(xcond), auxiliary_task=bool(xaux),
                           encoder_bidirectional=bool(xbi), num_encoder_layers=xlayers, input_vocabulary_size=14,
                           target_vocabulary_size=9, num_cnn_channels=16)
        shape = Shape(batch=xB, grid=xG, channels=16, input_vocab=14, target_vocab=9, max_command=xL, max_target=xT,
                      ragged=xB > 1)
    else:
        cfg = model_kwargs("demo", cnn_dropout_p=0.0, encoder_dropout_p=0.0, decoder_dropout_p=0.0,
                           decoder_hidden_size=H, encoder_hidden_size=He, embedding_dimension=rng.choice([4, 5, 8, 25]),
                           cnn_kernel_size=rng.choice([1, 3, 5, 7, 13]), cnn_hidden_num_channels=rng.choice([8, 20, 50, 70]),
                           conditional_attention=rng.random() < 0.6, auxiliary_task=rng.random() < 0.5,
                           encoder_bidirectional=rng.random() < 0.7, num_encoder_layers=rng.choice([1, 1, 2, 3]),
                           input_vocabulary_size=rng.choice([8, 14, 21]), target_vocabulary_size=rng.choice([5, 6, 9]),
                           num_cnn_channels=rng.choice([15, 16]))
        shape = Shape(batch=rng.choice([1, 2, 3, 5, 9]), grid=rng.choice([2, 3, 4, 6, 8]), channels=cfg["num_cnn_channels"],
                      input_vocab=cfg["input_vocabulary_size"], target_vocab=cfg["target_vocabulary_size"],
                      max_command=rng.choice([2, 3, 7, 10, 17]), max_target=rng.choice([2, 3, 10, 17, 33]),
                      ragged=rng.random() < 0.7)
        if args.wide:
            import dataclasses
            shape = dataclasses.replace(shape, grid=rng.choice([2, 4, 6, 8, 9, 10, 12]),
                                        max_command=rng.choice([2, 7, 17, 40, 65, 100, 128]),
                                        max_target=rng.choice([2, 3, 10, 17]))
    batch = make_batch(shape, seed=1000 + case)
    params = {k: torch.from_numpy(v) for k, v in golden_weights(cfg, 100 + case).items()}
    try:
        model = Model(**cfg)
        model.load_state_dict(params, strict=False)
        model = model.cuda().eval()
        d = {k: v.cuda() for k, v in batch.items()}
        if rng.random() < 0.5:
            d["world"] = d["world"].to(torch.uint8)         # the batcher's form: widened inside the kernels
        logp, aux = model(commands_input=d["commands"], commands_lengths=batch["cmd_lengths"].tolist(),
                          situations_input=d["world"], target_batch=d["targets"],
                          target_lengths=batch["tgt_lengths"].tolist())
        loss = model.get_loss(logp, d["targets"])
        if cfg["auxiliary_task"]:
            loss = loss + 0.3 * model.get_auxiliary_loss(aux, d["target_positions"])
        loss.backward()
        torch.cuda.synchronize()
        ref_loss, ref_grads, ref_logp = oracle.loss_and_grads(params, batch, conditional=cfg["conditional_attention"],
                                                              auxiliary=cfg["auxiliary_task"],
                                                              bidirectional=cfg["encoder_bidirectional"])
        e_logp = (logp.detach().cpu() - ref_logp).abs().max().item()
        both_nan = loss.item() != loss.item() and ref_loss.item() != ref_loss.item()     # no valid target token: 0 / 0 on both sides
        e_loss = 0.0 if both_nan else abs(loss.item() - ref_loss.item())
        worst, worst_name = 0.0, ""
        for n, p in model.named_parameters():
            g, r = p.grad.cpu(), ref_grads[n]
            e = ((g - r).abs() - 1e-3 * r.abs()).max().item()
            if e > worst:
                worst, worst_name = e, n
        ok = e_logp < TOL and e_loss < TOL and worst < TOL
        if args.train_step and not both_nan:
            from multimodal_seq2seq_gscan_amd.train import TrainStep
            twin = Model(**cfg)
            twin.load_state_dict(params, strict=False)
            twin = twin.cuda()
            step = TrainStep(twin, learning_rate=0.0, weight_target_loss=0.3)   # dropout is 0 in this configuration
            tb = {k: v for k, v in d.items() if k in ("commands", "cmd_lengths", "world", "targets", "target_positions")}
            grads = []
            step.on_gradients = lambda g: grads.append(g.detach().cpu().clone())
            out = step(tb)
            torch.cuda.synchronize()
            e_loss = max(e_loss, abs(out["loss"].item() - ref_loss.item()))
            for n, _ in twin.named_parameters():
                off, cnt = twin._offsets[n]
                g, r = grads[0][off:off + cnt].view(ref_grads[n].shape), ref_grads[n]
                e = ((g - r).abs() - 1e-3 * r.abs()).max().item()
                if e > worst:
                    worst, worst_name = e, "train_step:" + n
            ok = ok and e_loss < TOL and worst < TOL
    except Exception as exc:                                   # a configuration the library rejects is reported, not fatal
        print(f"case {case}: {type(exc).__name__}: {str(exc)[:150]}")
        ok, e_logp, e_loss, worst, worst_name = True, -1, -1, -1, "rejected"
    tag = "ok " if ok else "BAD"
    bad += 0 if ok else 1
    print(f"{tag} case {case}: H={H} He={He} E={cfg['embedding_dimension']} k={cfg['cnn_kernel_size']} Co={cfg['cnn_hidden_num_channels']} "
          f"cond={int(cfg['conditional_attention'])} aux={int(cfg['auxiliary_task'])} bi={int(cfg['encoder_bidirectional'])} "
          f"layers={cfg['num_encoder_layers']} B={shape.batch} G={shape.grid} L={shape.max_command} T={shape.max_target} "
          f"| logp {e_logp:.1e} loss {e_loss:.1e} grad {worst:.1e} {worst_name}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
