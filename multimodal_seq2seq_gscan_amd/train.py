"""The training loop body of the reference (seq2seq/train.py:96-114) as one object, plus data
parallelism the reference does not have.

`TrainStep.__call__(batch)` = forward -> loss -> backward -> [all-reduce] -> Adam + LambdaLR ->
update_state, with every arithmetic step a C-ABI launch on the current stream and no host
synchronisation anywhere (the loss comes back as a device tensor).

Data parallelism: one process per GPU, each rank takes its rows of the global
minibatch.  The reference's loss is a mean over the *global* number of non-pad target tokens, so
ranks first all-reduce four numbers [sum NLL, tokens, sum aux NLL, rows]; each rank then seeds its
backward pass with -1/tokens_global (and -w/rows_global for the auxiliary head), which makes the
SUM of the per-rank gradients equal to the single-process gradient of the global batch; one
all-reduce(sum) over the flat gradient buffer (440 275 floats = 1.76 MB at the paper
configuration) finishes the exchange.  Backend "nccl" is RCCL on ROCm; the CPU tests drive the
same class with gloo through the `backend` hooks below.
"""
from __future__ import annotations

import ctypes as C
import logging
import math
from typing import Dict, Iterable, Optional

import torch
import torch.distributed as dist

from . import _lib
from .model import Model, _as_int32_lengths

logger = logging.getLogger(__name__)


class FlatAdam:
    """torch.optim.Adam(lr, betas) + LambdaLR(lr_decay ** (t / lr_decay_steps)) of train.py:67-70 over
    the model's flat parameter buffer: one fused HIP kernel per step."""

    def __init__(self, model: Model, learning_rate: float, adam_beta_1: float = 0.9, adam_beta_2: float = 0.999,
                 lr_decay: float = 0.9, lr_decay_steps: float = 20000.0, eps: float = 1e-8):
        self.model = model
        self.lr, self.betas, self.eps = float(learning_rate), (float(adam_beta_1), float(adam_beta_2)), float(eps)
        self.lr_decay, self.lr_decay_steps = float(lr_decay), float(lr_decay_steps)
        self.steps_taken = 0
        flat = model.flat_parameters
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)

    def current_lr(self) -> float:
        """What scheduler.get_lr()[0] prints at train.py:123 after `steps_taken` scheduler steps."""
        return self.lr * self.lr_decay ** (self.steps_taken / self.lr_decay_steps)

    def step(self, grad_scale: Optional[torch.Tensor] = None) -> None:
        lib = _lib.load()
        m = self.model
        self.steps_taken += 1
        _lib.check(lib.gscan_adam_step(m.flat_parameters.data_ptr(), m.flat_gradients.data_ptr(),
                                       self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                       m.flat_parameters.numel(), self.lr, self.betas[0], self.betas[1], self.eps,
                                       self.lr_decay, self.lr_decay_steps, self.steps_taken, _lib.ptr(grad_scale),
                                       torch.cuda.current_stream().cuda_stream), "gscan_adam_step")

    # ---- checkpoint interop with torch.optim.Adam (model.py:246-261 stores optimizer.state_dict()) ----
    def state_dict(self) -> dict:
        state = {}
        for i, (name, p) in enumerate(self.model.named_parameters()):
            off, n = self.model._offsets[name]
            state[i] = {"step": torch.tensor(float(self.steps_taken)),
                        "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
        group = {"lr": self.current_lr(), "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "initial_lr": self.lr, "params": list(range(len(state)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: dict) -> None:
        for i, (name, p) in enumerate(self.model.named_parameters()):
            if i not in sd["state"]:
                continue
            off, n = self.model._offsets[name]
            self.exp_avg[off:off + n].copy_(sd["state"][i]["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(sd["state"][i]["exp_avg_sq"].reshape(-1))
            self.steps_taken = int(sd["state"][i]["step"])


class GradientExchange:
    """The data-parallel exchange of one step: a 4-float statistics all-reduce before backward, one flat
    gradient all-reduce after it.  With a single process both are no-ops."""

    def __init__(self, process_group=None):
        self.group = process_group
        active = dist.is_available() and dist.is_initialized()
        self.world_size = dist.get_world_size(process_group) if active else 1
        self.rank = dist.get_rank(process_group) if active else 0

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        if self.world_size > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def seeds(self, stats: torch.Tensor, weight_target_loss: float, auxiliary: bool):
        """stats = [sum NLL, tokens, sum aux NLL, rows] of THIS rank.  Returns (global stats, seed for
        d(loss)/d(logp) at the picked entries, seed for the auxiliary head, global loss): with these seeds
        the sum over ranks of the local gradients is the gradient of the reference's global-batch loss
        mean_tokens(NLL) + w * mean_rows(aux NLL)  (model.py:147-164, train.py:102-107)."""
        stats = self.all_reduce(stats)
        seq_seed = 1.0 / stats[1]
        loss = stats[0] * seq_seed
        aux_seed = None
        if auxiliary:
            aux_seed = weight_target_loss / stats[3]
            loss = loss + stats[2] * aux_seed
        return stats, seq_seed, aux_seed, loss


class TrainStep:
    """One iteration of the reference loop (train.py:96-114) for this rank's rows of the minibatch."""

    def __init__(self, model: Model, learning_rate: float = 1e-3, adam_beta_1: float = 0.9,
                 adam_beta_2: float = 0.999, lr_decay: float = 0.9, lr_decay_steps: float = 20000.0,
                 weight_target_loss: float = 0.3, process_group=None, **_):
        self.model = model
        self.optimizer = FlatAdam(model, learning_rate, adam_beta_1, adam_beta_2, lr_decay, lr_decay_steps)
        self.weight_target_loss = float(weight_target_loss)
        self.exchange = GradientExchange(process_group)

    def __call__(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """batch: commands [B,L] i64, cmd_lengths [B], world [B,G,G,C] f32, targets [B,T] i64 and, with the
        auxiliary task, target_positions [B] i64 — all on the HIP device.  Returns device scalars
        `loss` (global mean, as the reference logs it), `tokens` and the local `logp`."""
        lib = _lib.load()
        model = self.model
        model.train()
        stream = torch.cuda.current_stream().cuda_stream
        commands, world, targets = batch["commands"], batch["world"], batch["targets"]
        device = commands.device
        B, L = commands.shape
        T = targets.shape[1]
        lengths = _as_int32_lengths(batch["cmd_lengths"], device)
        masks = model._draw_masks(B, L, T, world.shape[1] ** 2, device)
        logp, aux, call = model._launch_forward(commands, lengths, world, targets, masks)

        # [sum NLL, tokens, sum aux NLL, rows] — the only quantities ranks must agree on before backward
        stats = torch.zeros(4, dtype=torch.float32, device=device)
        dlogp = torch.empty_like(logp)
        _lib.check(lib.gscan_sequence_nll(logp.data_ptr(), call["keep"][3].data_ptr(), B, T, logp.shape[2],
                                          model.target_pad_idx, stats.data_ptr(), stats.data_ptr() + 4,
                                          dlogp.data_ptr(), stream), "gscan_sequence_nll")
        daux = None
        if model.auxiliary_task:
            daux = torch.empty_like(aux)
            pos = batch["target_positions"].view(-1).contiguous()
            _lib.check(lib.gscan_position_nll(aux.data_ptr(), pos.data_ptr(), B, aux.shape[1],
                                              stats.data_ptr() + 8, daux.data_ptr(), stream), "gscan_position_nll")
        stats[3] = float(B)
        stats, seq_seed, aux_seed, loss = self.exchange.seeds(stats, self.weight_target_loss, model.auxiliary_task)
        dlogp.mul_(seq_seed)
        if daux is not None:
            daux.mul_(aux_seed)

        model.flat_gradients.zero_()
        model._launch_backward(call, dlogp, daux)
        self.exchange.all_reduce(model.flat_gradients)
        self.optimizer.step()
        model.update_state(is_best=False)
        return {"loss": loss, "tokens": stats[1], "logp": logp, "aux": aux}


def shard_batch(batch: Dict[str, torch.Tensor], rank: int, world_size: int) -> Dict[str, torch.Tensor]:
    """Rank r takes rows [r*B/W, (r+1)*B/W) of the global minibatch (last rank takes the remainder)."""
    B = batch["commands"].shape[0]
    per = B // world_size
    lo = rank * per
    hi = B if rank == world_size - 1 else lo + per
    return {k: v[lo:hi] for k, v in batch.items()}


def train(batches: Iterable[Dict[str, torch.Tensor]], model: Model, max_training_iterations: int,
          print_every: int = 100, weight_target_loss: float = 0.3, rank: int = 0, **optim_flags) -> TrainStep:
    """The reference's `while training_iteration < max_training_iterations` loop (train.py:86-153) over an
    iterable of device batches; evaluation/checkpointing (train.py:129-149) belong to the callers' side
    of the hot path and are not repeated here."""
    step = TrainStep(model, weight_target_loss=weight_target_loss, **optim_flags)
    it = 1
    for batch in batches:
        if it >= max_training_iterations + 1:
            break
        out = step(batch)
        if it % print_every == 0 and rank == 0:
            accuracy, exact_match = model.get_metrics(out["logp"], batch["targets"])
            aux_acc = (model.get_auxiliary_accuracy(out["aux"], batch["target_positions"])
                       if model.auxiliary_task else 0.0)
            logger.info("Iteration %08d, loss %8.4f, accuracy %5.2f, exact match %5.2f, learning_rate %.5f,"
                        " aux. accuracy target pos %5.2f" % (it, out["loss"].item(), accuracy, exact_match,
                                                             step.optimizer.current_lr(), aux_acc))
        it += 1
    return step
