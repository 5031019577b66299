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
import os
import math
from typing import Dict, Iterable, Optional

import torch
import torch.distributed as dist

from . import _lib
from .model import Model, _as_int32_lengths

logger = logging.getLogger(__name__)


class FlatAdam:
    """torch.optim.Adam(lr, betas) + LambdaLR(lr_decay ** (t / lr_decay_steps)) of train.py:67-70 over
    the model's flat parameter buffer: one fused HIP kernel per step, which also clears the gradient buffer
    (optimizer.zero_grad(), train.py:113).  The step-dependent scalars are kernel arguments."""

    def __init__(self, model: Model, learning_rate: float, adam_beta_1: float = 0.9, adam_beta_2: float = 0.999,
                 lr_decay: float = 0.9, lr_decay_steps: float = 20000.0, eps: float = 1e-8):
        self.model = model
        self.lr, self.betas, self.eps = float(learning_rate), (float(adam_beta_1), float(adam_beta_2)), float(eps)
        self.lr_decay, self.lr_decay_steps = float(lr_decay), float(lr_decay_steps)
        self.steps_taken = 0          # Adam's `step` (bias corrections); restored from a checkpoint
        self._resume_lr = None        # the rate of the first optimizer.step() after load_state_dict (see _lr_of_this_step)
        self.lr_steps = 0             # LambdaLR's counter; NOT restored on resume, as in the reference (train.py:68-84:
                                      # the scheduler is built before optimizer.load_state_dict and is not checkpointed,
                                      # so a resumed run starts again from the initial learning rate)
        flat = model.flat_parameters
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)

    def current_lr(self) -> float:
        """What scheduler.get_lr()[0] prints at train.py:123 after `steps_taken` scheduler steps."""
        return self.lr * self.lr_decay ** (self.lr_steps / self.lr_decay_steps)

    def advance(self) -> None:
        """One more optimizer.step() + scheduler.step() (train.py:111-112)."""
        self.steps_taken += 1
        self.lr_steps += 1

    def _lr_of_this_step(self) -> float:
        """The learning rate optimizer.step() number `lr_steps` runs with: the scheduler has stepped lr_steps-1 times.
        The FIRST step after a resume runs with the checkpoint's own rate: the reference's optimizer.load_state_dict
        overwrites param_groups[0]['lr'] behind the freshly built scheduler (train.py:68-84), whose next step() then
        puts the schedule back on its restarted curve."""
        if self._resume_lr is not None and self.lr_steps == 1:
            return self._resume_lr
        return self.lr * self.lr_decay ** ((self.lr_steps - 1) / self.lr_decay_steps)

    def launch_with_next_masks(self, count: Optional[torch.Tensor]) -> bool:
        """Adam + zero_grad (divided by `count` when given) AND the dropout masks of the next step in one launch
        (`gscan_adam_step_masks`), for a next batch of the shape the model saw last: the masks depend on a counter
        only, so the launch at the head of the next step disappears.  Returns False (nothing launched) when there
        are no device-drawn masks to draw ahead (eval / p = 0 / host masks / deeper encoders)."""
        m = self.model
        if (m._mask_key is None or m._mask_buffer is None or m._host_masks is not None or not m.training
                or max(m.dropout_p) <= 0.0 or m._hyper["NL"] > 1):
            return False
        lib = _lib.load()
        sizes = m._mask_key
        _lib.check(lib.gscan_adam_step_masks(m.flat_parameters.data_ptr(), m.flat_gradients.data_ptr(),
                                             self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                             m.flat_parameters.numel(), self._lr_of_this_step(), self.betas[0], self.betas[1],
                                             self.eps, 1.0, 1.0, self.steps_taken, _lib.ptr(count), m._mask_buffer.data_ptr(),
                                             sizes[0], sizes[1], sizes[2], m.dropout_p[0], m.dropout_p[1], m.dropout_p[2],
                                             m._dropout_seed, m._philox_stream(), torch.cuda.current_stream().cuda_stream),
                   "gscan_adam_step_masks")
        m._predrawn = (tuple(sizes), m._philox_stream(), m._mask_buffer.device)
        return True

    def launch_mean(self, count: torch.Tensor) -> None:
        """Adam + zero_grad on gradients of a SUM loss: divided by the device scalar `count` (global token count)."""
        lib = _lib.load()
        m = self.model
        _lib.check(lib.gscan_adam_step_mean(m.flat_parameters.data_ptr(), m.flat_gradients.data_ptr(),
                                            self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                            m.flat_parameters.numel(), self._lr_of_this_step(), self.betas[0],
                                            self.betas[1], self.eps, 1.0, 1.0, self.steps_taken, count.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream), "gscan_adam_step_mean")

    def launch(self, zero_grad: bool = True) -> None:
        lib = _lib.load()
        m = self.model
        fn = lib.gscan_adam_step_zero_grad if zero_grad else lib.gscan_adam_step
        _lib.check(fn(m.flat_parameters.data_ptr(), m.flat_gradients.data_ptr(), self.exp_avg.data_ptr(),
                      self.exp_avg_sq.data_ptr(), m.flat_parameters.numel(), self._lr_of_this_step(), self.betas[0],
                      self.betas[1], self.eps, 1.0, 1.0, self.steps_taken, None,
                      torch.cuda.current_stream().cuda_stream), "gscan_adam_step")

    def step(self, zero_grad: bool = True) -> None:
        self.advance()
        self.launch(zero_grad)

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
        self.lr_steps = 0
        groups = sd.get("param_groups") or [{}]
        self._resume_lr = float(groups[0]["lr"]) if "lr" in groups[0] else None


class RcclCommunicator:
    """One RCCL communicator of this process, driven through the C ABI (`gscan_comm_*`, `gscan_allreduce_f32`): the
    all-reduce is enqueued on the CALLER'S stream, right behind the backward kernels and in front of the optimiser —
    torch.distributed runs its collectives on a stream of its own, two event hops (~50 us) per step.  The 128-byte
    unique id travels from rank 0 to the others through the process group that is already up (any backend);
    `torch.cuda.set_device(local_rank)` must have happened before."""

    def __init__(self, process_group=None):
        lib = _lib.load()
        self.rank = dist.get_rank(process_group)
        self.world_size = dist.get_world_size(process_group)
        self._handle = C.c_void_p()
        # The transport decision is COLLECTIVE, and it is taken BEFORE anyone enters ncclCommInitRank (which blocks until
        # every rank has joined): (1) every rank probes that it can load RCCL at all (gscan_comm_available: dlopen and
        # symbols, no device call); (2) rank 0 ALWAYS broadcasts — the unique id, or None when it could not draw one;
        # (3) every rank contributes "RCCL loadable here AND an id arrived" to a MIN all-reduce over the process group
        # that is already up.  Only when every rank voted yes does anyone call gscan_comm_init; otherwise all of them
        # raise (GradientExchange logs it and all fall back to torch.distributed together).  A second vote behind the
        # init catches a rank whose init failed after joining (the others came back from ncclCommInitRank with it).
        # Round 6: what can fail LOCALLY on the way into ncclCommInitRank is checked in front of the first vote
        # too — the device this rank was told to use answers (a context exists, a kernel-less round trip completes) — and
        # the init itself runs under a watchdog: a rank still inside it after GSCAN_COMM_INIT_TIMEOUT seconds (default
        # 300; 0 = none) ends its PROCESS with a non-zero code (no re-exec; the launcher then ends the other ranks,
        # bench.py polls all of them), instead of every rank waiting forever for one that died on the way in.
        failure = None
        try:
            _lib.check(lib.gscan_comm_available(), "gscan_comm_available")
            self._probe_device()
        except (_lib.GscanError, RuntimeError) as e:
            failure = str(e)
        uid = C.create_string_buffer(_lib.COMM_ID_BYTES)
        box = [None]
        if self.rank == 0 and failure is None:
            try:
                _lib.check(lib.gscan_comm_unique_id(C.addressof(uid)), "gscan_comm_unique_id")
                box = [bytes(uid.raw)]
            except _lib.GscanError as e:
                failure = str(e)
        dist.broadcast_object_list(box, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                                   group=process_group)
        if box[0] is None:
            failure = failure or "rank 0 could not draw an RCCL unique id"
        if not self._vote(failure is None, process_group):
            raise RuntimeError(failure or "another rank cannot load RCCL or received no unique id")
        uid = C.create_string_buffer(box[0], _lib.COMM_ID_BYTES)
        watchdog = self._init_watchdog(self.rank)
        try:
            _lib.check(lib.gscan_comm_init(C.byref(self._handle), self.world_size, self.rank, C.addressof(uid)),
                       "gscan_comm_init")
        except _lib.GscanError as e:
            failure = str(e)
            self._handle = C.c_void_p()
        finally:
            if watchdog is not None:
                watchdog.cancel()
        if not self._vote(failure is None, process_group):
            self.close()
            raise RuntimeError(failure or "another rank could not create its RCCL communicator")

    @staticmethod
    def _probe_device() -> None:
        """The device-side preconditions of the collective init, checked where a failure is still a vote: the current
        device exists and completes a round trip (raises RuntimeError otherwise)."""
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device is visible to this rank")
        dev = torch.cuda.current_device()
        if dev >= torch.cuda.device_count():
            raise RuntimeError(f"device {dev} selected, {torch.cuda.device_count()} visible")
        torch.zeros(1, device="cuda").add_(1.0)
        torch.cuda.synchronize()

    @staticmethod
    def _init_watchdog(rank: int):
        """A timer that ends this PROCESS if the collective init has not come back in time (None: disabled)."""
        import threading
        seconds = float(os.environ.get("GSCAN_COMM_INIT_TIMEOUT", "300"))
        if seconds <= 0:
            return None

        def expire():
            logger.error("rank %d: still inside ncclCommInitRank after %.0f s (a peer never joined?): exiting", rank, seconds)
            logging.shutdown()
            os._exit(3)

        timer = threading.Timer(seconds, expire)
        timer.daemon = True
        timer.start()
        return timer

    @staticmethod
    def _vote(ok: bool, process_group) -> bool:
        """True when EVERY rank of the group passed True (a MIN all-reduce on the process group)."""
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32,
                            device="cuda" if dist.get_backend(process_group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=process_group)
        return int(flag.item()) == 1

    @property
    def nranks(self) -> int:
        """Ranks of the communicator as RCCL reports them (ncclCommCount), not as the process group does."""
        n = C.c_int(0)
        _lib.check(_lib.load().gscan_comm_count(self._handle, C.byref(n)), "gscan_comm_count")
        return int(n.value)

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks, on the current stream."""
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("RcclCommunicator.all_reduce takes a contiguous float32 device tensor")
        _lib.check(_lib.load().gscan_allreduce_f32(self._handle, t.data_ptr(), t.numel(),
                                                   torch.cuda.current_stream().cuda_stream), "gscan_allreduce_f32")
        return t

    def close(self) -> None:
        if self._handle:
            torch.cuda.synchronize()
            _lib.check(_lib.load().gscan_comm_destroy(self._handle), "gscan_comm_destroy")
            self._handle = C.c_void_p()


class GradientExchange:
    """The data-parallel exchange of one step.  Without the auxiliary loss: ONE all-reduce of
    [gradients of the local SUM loss | sum NLL, tokens, ., rows] (`mean_from_sums`).  With it (two different
    divisors): a 4-float statistics all-reduce before backward (`seeds`), one flat gradient all-reduce after it.
    With a single process all of them are no-ops.

    Transport of float32 device buffers when the process group's backend is "nccl" (= RCCL): the library's own
    communicator on the caller's stream (`RcclCommunicator`).  `native=False`, or a failure to create that
    communicator (logged), leaves them to torch.distributed — RCCL as well, through its communication stream."""

    def __init__(self, process_group=None, always_collective: bool = False, native: Optional[bool] = None,
                 buckets: Optional[int] = None):
        self.group = process_group
        # GSCAN_DP_BUCKETS=2 (round 6; default 1): the flat gradient crosses in two all-reduces — the group the backward
        # pass finishes ~45 us early (decoder, attentions, bridge: the tail of the flat buffer, with the loss statistics) on
        # a communication stream UNDER the rest of the pass, the rest (convolution kernels, command encoder) on the step's
        # stream behind it (`all_reduce_two_buckets`).
        self.buckets = int(os.environ.get("GSCAN_DP_BUCKETS", "1")) if buckets is None else int(buckets)
        self._comm_stream = None
        active = dist.is_available() and dist.is_initialized()
        self.world_size = dist.get_world_size(process_group) if active else 1
        self.rank = dist.get_rank(process_group) if active else 0
        # always_collective: issue the collectives even on a one-rank group (drives the real RCCL launch and its
        # stream hand-over on a single device; tests)
        self.collective = active and (self.world_size > 1 or always_collective)
        # gloo has no device collectives: device buffers are staged through host memory (two processes sharing one
        # GPU in the tests; RCCL refuses two ranks on one device).  "nccl" = RCCL reduces in place over xGMI.
        self.host_staged = self.collective and dist.get_backend(process_group) == "gloo"
        self.comm: Optional[RcclCommunicator] = None
        if native is None:
            native = os.environ.get("GSCAN_NATIVE_ALLREDUCE", "1") != "0"
        # two buckets: the early group travels on a communicator OF ITS OWN, bound to the communication stream for good — one
        # communicator used from two streams in turn makes RCCL order them against each other (measured on one rank with
        # the auxiliary head's three collectives per step, caller -> communication -> caller: 0.47 -> 0.77 ms per step)
        self.comm_early: Optional[RcclCommunicator] = None
        self._registered = None
        if self.collective and not self.host_staged and native and torch.cuda.is_available():
            try:
                self.comm = RcclCommunicator(process_group)
                if self.buckets == 2:
                    self.comm_early = RcclCommunicator(process_group)
            except (_lib.GscanError, RuntimeError) as e:      # raised on EVERY rank or on none (collective decision)
                logger.warning("RCCL communicator on the caller's stream unavailable (%s): gradients go through "
                               "torch.distributed's RCCL stream instead", e)
                if self.comm is not None:                     # the second one failed (on every rank): one bucket
                    self.buckets = 1

    def close(self) -> None:
        """Destroy the library's RCCL communicator (include/gscan_hip.h: gscan_comm_destroy at exit); call before
        dist.destroy_process_group()."""
        if getattr(self, "_registered", None) is not None:
            _lib.check(_lib.load().gscan_comm_set_early_allreduce(None, None, 0), "gscan_comm_set_early_allreduce")
            self._registered = None
        for name in ("comm_early", "comm"):
            comm = getattr(self, name)
            if comm is not None:
                comm.close()
                setattr(self, name, None)

    def all_reduce(self, t: torch.Tensor, comm: Optional[RcclCommunicator] = None) -> torch.Tensor:
        if not self.collective:
            return t
        comm = comm if comm is not None else self.comm
        if t.is_cuda and self.host_staged:
            host = t.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(host)
        elif comm is not None and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous():
            comm.all_reduce(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def register_early_range(self, t: torch.Tensor, split: int) -> bool:
        """Production form of the two-bucket exchange: hand the library the early group's communicator and range ONCE
        (gscan_comm_set_early_allreduce); every backward pass then all-reduces `t[split:]` itself on its first leaf stream,
        behind the last kernel that writes into it, and `all_reduce_two_buckets` only reduces `t[:split]`.  False (and
        nothing registered) when the early group has no communicator of its own (gloo / torch.distributed transports: the
        communication-stream form below is used)."""
        if self.comm_early is None or not self.collective or self.buckets != 2:
            return False
        early = t[split:]
        _lib.check(_lib.load().gscan_comm_set_early_allreduce(self.comm_early._handle, early.data_ptr(), early.numel()),
                   "gscan_comm_set_early_allreduce")
        self._registered = (t.data_ptr(), split, t.numel())
        return True

    def all_reduce_two_buckets(self, t: torch.Tensor, split: int, wait_early=None) -> torch.Tensor:
        """In-place sum of `t` over the ranks as TWO collectives: `t[split:]` — complete on the device as soon as
        `wait_early(stream)` lets a stream pass (gscan_early_gradients_wait: the backward pass's first leaf stream is
        done) — is reduced on a communication stream of its own while the producer of `t[:split]` still runs on the
        current stream; `t[:split]` follows on the current stream, which then waits for the communication stream.  Every
        rank issues the two collectives in the same order.  Same result as ONE all-reduce of `t` (elementwise sums)."""
        if not self.collective:
            return t
        if not t.is_cuda:                      # host tensors (CPU tests of the arithmetic): nothing to overlap
            self.all_reduce(t[split:])
            self.all_reduce(t[:split])
            return t
        if getattr(self, "_registered", None) is not None:
            # the backward pass has all-reduced the early range on its leaf stream (and joined it): the rest follows here
            base, at, total = self._registered
            if (t.data_ptr(), split) != (base, at) or t.numel() > total:
                raise ValueError("two-bucket exchange: the registered early range is of another buffer or split")
            self.all_reduce(t[:split])
            return t
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=t.device)
        caller, cs = torch.cuda.current_stream(), self._comm_stream
        if wait_early is not None:
            wait_early(cs.cuda_stream)
        else:
            cs.wait_stream(caller)             # no early point known: the communication stream starts behind the producer
        with torch.cuda.stream(cs):
            self.all_reduce(t[split:], comm=self.comm_early)
            done = cs.record_event()
        self.all_reduce(t[:split])
        caller.wait_event(done)
        return t

    def mean_from_sums(self, grads_and_stats: torch.Tensor, split: Optional[int] = None, wait_early=None):
        """grads_and_stats = [d(sum NLL of this rank's rows)/d(params) | sum NLL, tokens, sum aux NLL, rows]: after
        the all-reduce the reference's global-batch loss is stats[0]/stats[1] and its gradient is the reduced
        gradient divided by stats[1] (model.py:147-160: mean over the live tokens of the WHOLE minibatch).  Returns
        (reduced buffer, global token count as a 1-element view, loss)."""
        if self.buckets == 2 and split:
            self.all_reduce_two_buckets(grads_and_stats, split, wait_early)
        else:
            self.all_reduce(grads_and_stats)
        stats = grads_and_stats[-4:]
        return grads_and_stats, stats[1:2], stats[0] / stats[1]

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
    """One iteration of the reference loop (train.py:96-114) for this rank's rows of the minibatch.

    Launches are eager: replaying captured HIP graphs was built in round 2 and measured slower on this stack
    (DESIGN.md 6), and is gone."""

    def __init__(self, model: Model, learning_rate: float = 1e-3, adam_beta_1: float = 0.9,
                 adam_beta_2: float = 0.999, lr_decay: float = 0.9, lr_decay_steps: float = 20000.0,
                 weight_target_loss: float = 0.3, process_group=None,
                 fused_loss: Optional[bool] = None, single_exchange: Optional[bool] = None,
                 always_collective: bool = False, on_gradients=None,
                 native_allreduce: Optional[bool] = None, dp_buckets: Optional[int] = None, **ignored):
        if ignored.pop("graph", None):
            # hipGraph replay left the product path in round 4 (slower than eager launches on this stack)
            logger.warning("TrainStep(graph=True): graph replay is no longer available, the step launches eagerly")
        if ignored:                                          # the reference's callers pass their whole flag dict (model.py:26-32)
            logger.debug("TrainStep: ignoring arguments %s", sorted(ignored))
        self.model = model
        self.optimizer = FlatAdam(model, learning_rate, adam_beta_1, adam_beta_2, lr_decay, lr_decay_steps)
        self.weight_target_loss = float(weight_target_loss)
        self.exchange = GradientExchange(process_group, always_collective, native_allreduce, buckets=dp_buckets)
        # two-bucket exchange (GSCAN_DP_BUCKETS=2): the EARLY group of gradients is the tail of the flat buffer from the
        # bridge on (named_parameters() order: convolutions, visual attention, command encoder | bridge, textual attention,
        # decoder) plus the statistics behind it; the visual attention's 25 k floats are early too but sit between the two
        # late groups and travel with them.  gscan_early_gradients_wait is the device-side "early group complete" point.
        offsets = getattr(model, "_offsets", {})
        self._early_split = offsets["enc_hidden_to_dec_hidden.weight"][0] if "enc_hidden_to_dec_hidden.weight" in offsets else 0

        def wait_early(stream_handle):
            _lib.check(_lib.load().gscan_early_gradients_wait(stream_handle), "gscan_early_gradients_wait")
        self._wait_early = wait_early
        if self.exchange.buckets == 2 and self._early_split:
            # [gradients | 4 statistics]: the one-collective form reduces the statistics with the early group, the
            # two-collective form (auxiliary head) the gradients only — the registered range covers the larger one, and
            # summing the statistics slots again is harmless there (they are rewritten by every backward pass)
            self.exchange.register_early_range(model._grad_store, self._early_split)
        # every rank draws its own dropout masks (Philox streams keyed by seed, rank and step)
        model.set_dropout_rank(self.exchange.rank)
        # on_gradients(flat mean-loss gradient of the global batch): called between the exchange and the optimiser
        # (which clears the buffer); diagnostics and tests only — it costs a device pass in the one-collective form
        self.on_gradients = on_gradients
        # A single process needs no statistics exchange between forward and backward: the backward kernels seed
        # themselves from the loss partials of the forward pass (gscan_backward_nll), two launches fewer.
        self.fused_loss = (not self.exchange.collective) if fused_loss is None else bool(fused_loss)
        if self.fused_loss and self.exchange.world_size > 1:
            raise ValueError("fused_loss needs the global token count: not available with more than one process")
        # Several processes, no auxiliary loss: every rank back-propagates its SUM loss, the statistics ride behind
        # the gradients in one all-reduce and Adam divides by the global token count.
        if single_exchange is None:
            single_exchange = self.exchange.collective and not model.auxiliary_task
        elif single_exchange and model.auxiliary_task:
            raise ValueError("the one-collective step needs a single loss term")
        self.single_exchange = bool(single_exchange)
        if self.single_exchange:
            self.fused_loss = False
        # eager steps whose backward pass seeds itself (fused_loss, or the one-collective data-parallel form) make ONE
        # library call per iteration (gscan_train_step_nll; GSCAN_ONE_CALL=0: gscan_forward + gscan_backward_nll)
        self.one_call = os.environ.get("GSCAN_ONE_CALL", "1") != "0"
        device = model.flat_parameters.device
        self.stats = torch.zeros(4, dtype=torch.float32, device=device)
        self.seeds = torch.zeros(3, dtype=torch.float32, device=device)
        model.flat_gradients.zero_()
        model.attach_gradients(zero=False)

    # ---- the three launch sections of a step ------------------------------------------------
    def _section_forward(self, batch, train_nll=None) -> dict:
        lib = _lib.load()
        model = self.model
        stream = torch.cuda.current_stream().cuda_stream
        commands, world, targets = batch["commands"], batch["world"], batch["targets"]
        B, L = commands.shape
        T = targets.shape[1]
        masks = model._draw_masks(B, L, T, world.shape[1] ** 2, commands.device)
        pos = batch["target_positions"] if model.auxiliary_task else None
        if train_nll is not None:      # forward, loss and backward in ONE library call (gscan_train_step_nll)
            logp, aux, call = model._launch_forward(commands, batch["cmd_lengths"], world, targets, masks, pos,
                                                    train_nll=train_nll)
            return {"logp": logp, "aux": aux, "call": call}
        logp, aux, call = model._launch_forward(commands, batch["cmd_lengths"], world, targets, masks, pos)
        if self.fused_loss or self.single_exchange:
            return {"logp": logp, "aux": aux, "call": call}
        dlogp = torch.empty_like(logp)
        daux = torch.empty_like(aux) if model.auxiliary_task else None
        pos = batch["target_positions"].view(-1).contiguous() if model.auxiliary_task else None
        _lib.check(lib.gscan_step_losses(logp.data_ptr(), call["keep"][3].data_ptr(),
                                         aux.data_ptr() if model.auxiliary_task else None, _lib.ptr(pos), B, T,
                                         logp.shape[2], aux.shape[1] if model.auxiliary_task else 0,
                                         model.target_pad_idx, self.stats.data_ptr(), dlogp.data_ptr(),
                                         _lib.ptr(daux), stream), "gscan_step_losses")
        return {"logp": logp, "aux": aux, "call": call, "dlogp": dlogp, "daux": daux, "pos": pos}

    def _section_backward(self, fw: dict) -> None:
        lib = _lib.load()
        if self.fused_loss:
            self.model._launch_backward_nll(fw["call"], self.weight_target_loss, self.stats, self.seeds)
            return
        _lib.check(lib.gscan_loss_seeds(self.stats.data_ptr(), self.weight_target_loss,
                                        int(self.model.auxiliary_task), self.seeds.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream), "gscan_loss_seeds")
        self.model._launch_backward(fw["call"], fw["dlogp"], fw["daux"], seeds=self.seeds, attach=False)

    def _host_prologue(self) -> None:
        """Per-step host work: only the step counter moves (the scalars travel as kernel arguments)."""
        self.optimizer.advance()

    def _result(self, fw: dict) -> Dict[str, torch.Tensor]:
        self.model.update_state(is_best=False)
        return {"loss": self.seeds[2], "tokens": self.stats[1], "logp": fw["logp"], "aux": fw["aux"]}

    def close(self) -> None:
        """Release what the step holds outside torch's allocator: the library's RCCL communicator.  Call it before
        dist.destroy_process_group() / interpreter exit (bench.py and train_on_dataset do)."""
        self.exchange.close()

    # ---- execution ------------------------------------------------------------------------------
    def __call__(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """batch: commands [B,L] i64, cmd_lengths [B], world [B,G,G,C] f32, targets [B,T] i64 and, with the
        auxiliary task, target_positions [B] i64 — all on the HIP device.  Returns device scalars
        `loss` (global mean, as the reference logs it), `tokens` and the local `logp`."""
        model = self.model
        model.train()
        device = batch["commands"].device
        batch = dict(batch, cmd_lengths=_as_int32_lengths(batch["cmd_lengths"], device))
        self._host_prologue()
        if self.single_exchange:
            store = model._grad_store
            if self.one_call:
                fw = self._section_forward(batch, train_nll=(self.weight_target_loss, True, store[-4:], self.seeds))
            else:
                fw = self._section_forward(batch)
                model._launch_backward_nll(fw["call"], self.weight_target_loss, store[-4:], self.seeds, sum_reduction=True)
            _, count, loss = self.exchange.mean_from_sums(store, self._early_split, self._wait_early)
            if self.on_gradients is not None:
                self.on_gradients(model.flat_gradients / count)
            if not self.optimizer.launch_with_next_masks(count):
                self.optimizer.launch_mean(count)
            model.update_state(is_best=False)
            return {"loss": loss, "tokens": count[0], "logp": fw["logp"], "aux": fw["aux"]}
        if self.fused_loss and self.one_call:      # one process, eager: forward + loss + backward in one library call
            fw = self._section_forward(batch, train_nll=(self.weight_target_loss, False, self.stats, self.seeds))
        else:
            fw = self._section_forward(batch)
            self.exchange.all_reduce(self.stats)
            self._section_backward(fw)
        if self.exchange.buckets == 2 and self._early_split:
            self.exchange.all_reduce_two_buckets(model.flat_gradients, self._early_split, self._wait_early)
        else:
            self.exchange.all_reduce(model.flat_gradients)
        if self.on_gradients is not None:
            self.on_gradients(model.flat_gradients)
        if not self.optimizer.launch_with_next_masks(None):
            self.optimizer.launch(zero_grad=True)
        return self._result(fw)


def shard_batch(batch: Dict[str, torch.Tensor], rank: int, world_size: int) -> Dict[str, torch.Tensor]:
    """Rank r takes rows [floor(r*B/W), floor((r+1)*B/W)) of the global minibatch: shard sizes differ by at most
    one row.  Every rank needs at least one row (a rank without rows could not join the step's collectives and the
    others would wait for it forever): B < W raises on EVERY rank, since all ranks see the same B."""
    B = batch["commands"].shape[0]
    if B < world_size:
        raise ValueError(f"a global batch of {B} rows cannot be sharded over {world_size} ranks "
                         f"(train_on_dataset drops such a trailing batch)")
    lo, hi = rank * B // world_size, (rank + 1) * B // world_size
    return {k: v[lo:hi] for k, v in batch.items()}


def train(batches: Iterable[Dict[str, torch.Tensor]], model: Model, max_training_iterations: int,
          print_every: int = 100, weight_target_loss: float = 0.3, rank: int = 0, optimizer_state_dict=None,
          **optim_flags) -> TrainStep:
    """The reference's `while training_iteration < max_training_iterations` loop (train.py:86-153) over an
    iterable of device batches; evaluation/checkpointing (train.py:129-149) belong to the callers' side
    of the hot path and are not repeated here."""
    step = TrainStep(model, weight_target_loss=weight_target_loss, **optim_flags)
    if optimizer_state_dict is not None:
        step.optimizer.load_state_dict(optimizer_state_dict)       # train.py:82
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


def train_on_dataset(data_path: str, data_directory: str, generate_vocabularies: bool, input_vocab_path: str,
                     target_vocab_path: str, training_batch_size: int, max_training_iterations: int,
                     print_every: int, evaluate_every: int, max_decoding_steps: int, output_directory: str,
                     resume_from_file: str = "", k: int = 0, max_training_examples=None, max_testing_examples=None,
                     weight_target_loss: float = 0.3, seed: int = 42, rank: int = 0, world_size: int = 1,
                     evaluation_batch_size: int = 256, length_bucket_batches: int = 0, **flags) -> Model:
    """seq2seq/train.py:17-154 on a gSCAN dataset file: the reference's loop — shuffle, iterate batches, step,
    log every `print_every`, greedy-decode the dev split every `evaluate_every` and checkpoint on a new best exact
    match — with the HIP step, the packed batcher (dataset.py) and batched evaluation.  Under torch.distributed
    every rank reads the same file, shuffles with the same seed and trains on its rows of each batch."""
    import random
    import numpy as np
    from .config import model_kwargs
    from .dataset import BatchStager, GroundedScanDataset
    from .predict import evaluate_sums

    torch.manual_seed(seed)                                     # train.py:27
    np.random.seed(seed)                                        # the same shuffle on every rank
    random.seed(seed)                                           # the same k-shot sample on every rank (dataset.load_examples)
    logger.info("Loading Training set...")
    training_set = GroundedScanDataset(data_path, data_directory, split="train", input_vocabulary_file=input_vocab_path,
                                       target_vocabulary_file=target_vocab_path,
                                       generate_vocabulary=generate_vocabularies, k=k)
    training_set.read_dataset(max_examples=max_training_examples)
    logger.info("Done Loading Training set.")
    logger.info("  Loaded {} training examples.".format(training_set.num_examples))
    logger.info("  Input vocabulary size training set: {}".format(training_set.input_vocabulary_size))
    logger.info("  Most common input words: {}".format(training_set.input_vocabulary.most_common(5)))
    logger.info("  Output vocabulary size training set: {}".format(training_set.target_vocabulary_size))
    logger.info("  Most common target words: {}".format(training_set.target_vocabulary.most_common(5)))
    if generate_vocabularies and rank == 0:
        training_set.save_vocabularies(input_vocab_path, target_vocab_path)
        logger.info("Saved vocabularies to {} for input and {} for target.".format(input_vocab_path, target_vocab_path))
    logger.info("Loading Dev. set...")
    dev_set = GroundedScanDataset(data_path, data_directory, split="dev", k=0,
                                  vocabularies=(training_set.input_vocabulary, training_set.target_vocabulary))
    dev_set.read_dataset(max_examples=None)
    dev_set.shuffle_data()                                       # train.py:52-53
    logger.info("Done Loading Dev. set.")

    cfg = model_kwargs("compositional")
    cfg.update({key: flags[key] for key in cfg if key in flags})
    cfg.update(input_vocabulary_size=training_set.input_vocabulary_size,
               target_vocabulary_size=training_set.target_vocabulary_size,
               num_cnn_channels=training_set.image_channels,
               input_padding_idx=training_set.input_vocabulary.pad_idx,
               target_pad_idx=training_set.target_vocabulary.pad_idx,
               target_eos_idx=training_set.target_vocabulary.eos_idx, output_directory=output_directory)
    model = Model(**cfg).cuda()
    optim = {key: flags[key] for key in ("learning_rate", "adam_beta_1", "adam_beta_2", "lr_decay", "lr_decay_steps")
             if key in flags}
    step = TrainStep(model, weight_target_loss=weight_target_loss, **optim)
    best_exact_match = 0
    if resume_from_file:
        assert os.path.isfile(resume_from_file), "No checkpoint found at {}".format(resume_from_file)
        logger.info("Loading checkpoint from file at '{}'".format(resume_from_file))
        step.optimizer.load_state_dict(model.load_model(resume_from_file))
        logger.info("Loaded checkpoint '{}' (iter {})".format(resume_from_file, model.trained_iterations))
    logger.info("Training starts..")
    training_iteration = model.trained_iterations if resume_from_file else 1
    vocab = dev_set.target_vocabulary
    # one ring of pinned / device slabs for the whole run: a batch is gathered into a slab, crosses PCIe in one
    # asynchronous copy a batch ahead of the step, and stays uint8 (the kernels widen the world in registers)
    stager = BatchStager(model.flat_parameters.device,
                         training_set.slab_bytes(-(-training_batch_size // max(1, world_size))))
    keys = ("commands", "cmd_lengths", "world", "targets", "tgt_lengths", "target_positions")
    while training_iteration < max_training_iterations:          # train.py:88
        training_set.shuffle_data(bucket_batches=length_bucket_batches, batch_size=training_batch_size)
        # under data parallelism every rank stages only ITS rows of each global batch (1/W of the host gather and of
        # the PCIe traffic); the epoch's short trailing batch (gSCAN_dataset.py:195-196) is dropped by every rank when
        # it holds fewer rows than there are ranks.  NOTE: a shard is padded to ITS longest rows, which changes
        # nothing in the loss (padding is inert) and saves decoder steps on ranks with short rows.
        for staged in training_set.batches(training_batch_size, stager=stager, row_shard=(rank, world_size)):
            batch = {key: staged[key] for key in keys}
            out = step(batch)
            if training_iteration % print_every == 0 and rank == 0:
                accuracy, exact_match = model.get_metrics(out["logp"], batch["targets"])
                aux_acc = (model.get_auxiliary_accuracy(out["aux"], batch["target_positions"])
                           if model.auxiliary_task else 0.0)
                logger.info("Iteration %08d, loss %8.4f, accuracy %5.2f, exact match %5.2f, learning_rate %.5f,"
                            " aux. accuracy target pos %5.2f" % (training_iteration, out["loss"].item(), accuracy,
                                                                 exact_match, step.optimizer.current_lr(), aux_acc))
            if training_iteration % evaluate_every == 0:                    # train.py:129-149
                # Every rank decodes its share of the dev batches and the four sums are all-reduced, so no rank sits
                # in the next step's collective while rank 0 decodes the whole split (and all ranks agree on "best").
                if rank == 0:
                    logger.info("Evaluating..")
                limit = max_testing_examples and -(-max_testing_examples // world_size)
                sums = torch.tensor(evaluate_sums(
                    dev_set.get_data_iterator(batch_size=evaluation_batch_size, shard=(rank, world_size),
                                              world_dtype=torch.uint8), model=model,
                    max_decoding_steps=max_decoding_steps, pad_idx=vocab.pad_idx, sos_idx=vocab.sos_idx,
                    eos_idx=vocab.eos_idx, max_examples_to_evaluate=limit), dtype=torch.float64)
                if world_size > 1:
                    sums = step.exchange.all_reduce(sums.cuda()).cpu()
                n = max(float(sums[3]), 1.0)
                accuracy, exact_match, target_accuracy = (float(sums[0]) / n, 100.0 * float(sums[1]) / n,
                                                          float(sums[2]) / n)
                if rank == 0:
                    logger.info("  Evaluation Accuracy: %5.2f Exact Match: %5.2f "
                                " Target Accuracy: %5.2f" % (accuracy, exact_match, target_accuracy))
                if exact_match > best_exact_match:
                    best_exact_match = exact_match
                    model.update_state(accuracy=accuracy, exact_match=exact_match, is_best=True)
                    if rank == 0:
                        model.save_checkpoint(file_name="checkpoint.pth.tar", is_best=True,
                                              optimizer_state_dict=step.optimizer.state_dict())
            training_iteration += 1
            if training_iteration > max_training_iterations:
                break
    step.close()                                                 # the library's RCCL communicator, before the process group goes
    logger.info("Finished training.")
    return model
