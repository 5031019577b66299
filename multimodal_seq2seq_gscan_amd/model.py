"""`Model`: the reference's model facade (seq2seq/model.py:23-261) on top of libgscan_hip.so.

Same constructor keywords, methods, attributes and `state_dict` keys as the reference class, so
`seq2seq/train.py` and checkpoints written by the reference work unchanged.  What differs is
everything underneath: there are no torch.nn forward passes.  The parameter tree below only
*holds* tensors (created by the same torch.nn constructors in the same order, hence identical
seeded initialisation, train.py:27,58-64); all arithmetic is two C-ABI calls per step —
`gscan_forward` and `gscan_backward` — operating on ONE flat fp32 parameter buffer and ONE flat
gradient buffer that every named parameter / `.grad` is a view of.  The flat buffers are what
the data-parallel all-reduce and the fused Adam kernel consume (train.py in this package).

There is deliberately no CPU path here: CPU tensors raise.  The CPU restatement used for parity
checks lives in `oracle/` and is never imported by this package.
"""
from __future__ import annotations

import ctypes as C
import logging
import os
import shutil
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib

logger = logging.getLogger(__name__)


# ------------------------------------------------------------------------------------------
# parameter holders: attribute names = the reference's module names (checkpoint ABI)
# ------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    def forward(self, *args, **kwargs):  # pragma: no cover - arithmetic lives in the HIP library
        raise RuntimeError("parameter holder: the forward pass is Model.forward (HIP)")


class _HipLinear(nn.Linear):
    """An nn.Linear (same parameters, same initialisation, same state_dict keys) whose forward is the library's
    fp32 MFMA GEMM.  predict.py:87-96 calls `visual_attention.key_layer`, `textual_attention.key_layer` and
    `enc_hidden_to_dec_hidden` directly, so these three layers must compute; inference only (no autograd)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("this layer runs on the HIP device only (there is no CPU fallback in this package)")
        lib = _lib.load()
        xin = x.detach().to(torch.float32).contiguous()
        K, N = self.in_features, self.out_features
        if xin.shape[-1] != K:
            raise ValueError(f"expected {K} input features, got {xin.shape[-1]}")
        M = xin.numel() // K
        out = torch.empty(*xin.shape[:-1], N, dtype=torch.float32, device=x.device)
        _lib.check(lib.gscan_gemm_f32(M, N, K, 1.0, xin.data_ptr(), K, 1, self.weight.data_ptr(), 1, K, 0.0,
                                      out.data_ptr(), N, _lib.ptr(self.bias), 0, None, 1,
                                      torch.cuda.current_stream().cuda_stream), "gscan_gemm_f32")
        return out


class _WorldEncoderParams(_Holder):
    """cnn_model.py:7-20 — three same-padded convolutions with kernels 1, 5 and cnn_kernel_size."""

    def __init__(self, channels: int, hidden_channels: int, kernel_size: int):
        super().__init__()
        self.conv_1 = nn.Conv2d(channels, hidden_channels, kernel_size=1, padding=0)
        self.conv_2 = nn.Conv2d(channels, hidden_channels, kernel_size=5, padding=2)
        self.conv_3 = nn.Conv2d(channels, hidden_channels, kernel_size=kernel_size, padding=kernel_size // 2)
        self.output_dimension = hidden_channels * 3


class _AttentionParams(_Holder):
    """seq2seq_model.py:99-103 — three bias-free projections."""

    def __init__(self, key_size: int, query_size: int, hidden_size: int):
        super().__init__()
        self.key_layer = _HipLinear(key_size, hidden_size, bias=False)
        self.query_layer = nn.Linear(query_size, hidden_size, bias=False)
        self.energy_layer = nn.Linear(hidden_size, 1, bias=False)


class _CommandEncoderParams(_Holder):
    """seq2seq_model.py:26-45"""

    def __init__(self, vocab: int, embedding_dim: int, hidden_size: int, num_layers: int, bidirectional: bool,
                 padding_idx: int):
        super().__init__()
        self.embedding = nn.Embedding(vocab, embedding_dim, padding_idx=padding_idx)
        self.lstm = nn.LSTM(input_size=embedding_dim, hidden_size=hidden_size, num_layers=num_layers,
                            bidirectional=bidirectional)


class _DecoderParams(_Holder):
    """seq2seq_model.py:333-357; shares the two attention holders with the model (":356-357")."""

    def __init__(self, hidden_size: int, output_size: int, num_layers: int, padding_idx: int,
                 textual_attention: _AttentionParams, visual_attention: _AttentionParams, conditional: bool):
        super().__init__()
        if conditional:
            self.queries_to_keys = nn.Linear(hidden_size * 2, hidden_size)
        self.embedding = nn.Embedding(output_size, hidden_size, padding_idx=padding_idx)
        self.lstm = nn.LSTM(hidden_size * 3, hidden_size, num_layers=num_layers)
        self.textual_attention = textual_attention
        self.visual_attention = visual_attention
        self.output_to_hidden = nn.Linear(hidden_size * 4, hidden_size, bias=False)
        self.hidden_to_output = nn.Linear(hidden_size, output_size, bias=False)
        self.num_layers = num_layers

    def initialize_hidden(self, encoder_message: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """seq2seq_model.py:494-504: hidden and cell state of every layer start from the same message."""
        message = encoder_message.unsqueeze(0).expand(self.num_layers, -1, -1).contiguous()
        return message.clone(), message.clone()


def _dropout_in_kernel() -> bool:
    """GSCAN_DROPOUT_IN_KERNEL (default 1): training-mode dropout is drawn inside the kernels that apply it; 0: masks in
    memory (round 4's path, kept for A/B runs)."""
    return os.environ.get("GSCAN_DROPOUT_IN_KERNEL", "1") != "0"


def _as_int32_lengths(lengths, device) -> torch.Tensor:
    """Lengths arrive as lists, numpy float64 arrays (gSCAN_dataset.py:276) or tensors."""
    if isinstance(lengths, torch.Tensor):
        return lengths.to(device=device, dtype=torch.int32)
    return torch.tensor([int(v) for v in lengths], dtype=torch.int32).to(device, non_blocking=True)


def _world_operand(world: torch.Tensor) -> torch.Tensor:
    """The world tensor as the kernels read it: uint8 stays uint8 (Grid.encode's own dtype, minigrid.py:384; the
    batcher ships it that way, 1 byte per element, and the convolution kernels widen it in registers), everything
    else becomes float32 as in the reference (gSCAN_dataset.py:262-264)."""
    if world.dtype == torch.uint8:
        return world.contiguous()
    return world.to(torch.float32).contiguous()


# ------------------------------------------------------------------------------------------
# autograd glue: one node for the whole network, one for each loss
# ------------------------------------------------------------------------------------------
class _NetworkFunction(torch.autograd.Function):
    """logp, aux = f(parameters, batch).  Parameter gradients do not travel through autograd:
    backward() adds them straight into the model's flat gradient buffer (which every
    `param.grad` is a view of); the `anchor` input only makes autograd call us."""

    @staticmethod
    def forward(ctx, anchor, model, commands, lengths, world, targets, masks):
        logp, aux, call = model._launch_forward(commands, lengths, world, targets, masks)
        ctx.model, ctx.call = model, call
        return logp, aux

    @staticmethod
    def backward(ctx, dlogp, daux):
        ctx.model._launch_backward(ctx.call, dlogp, daux if ctx.call["dims"].auxiliary else None)
        return (None,) * 7


class _SequenceNLL(torch.autograd.Function):
    """Model.get_loss (model.py:147-160)."""

    @staticmethod
    def forward(ctx, logp, targets, pad):
        lib = _lib.load()
        B, T, V = logp.shape
        logp = logp.contiguous()
        out = torch.empty(2, dtype=torch.float32, device=logp.device)
        dlogp = torch.empty_like(logp)
        _lib.check(lib.gscan_sequence_nll(logp.data_ptr(), targets.data_ptr(), B, T, V, pad, out.data_ptr(),
                                          out.data_ptr() + 4, dlogp.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "gscan_sequence_nll")
        ctx.save_for_backward(dlogp, out)
        return out[0] / out[1]

    @staticmethod
    def backward(ctx, g):
        dlogp, out = ctx.saved_tensors
        return dlogp * (g / out[1]), None, None


class _PositionNLL(torch.autograd.Function):
    """Model.get_auxiliary_loss (model.py:162-164): mean over the batch."""

    @staticmethod
    def forward(ctx, aux_logp, positions):
        lib = _lib.load()
        B, M = aux_logp.shape
        aux_logp = aux_logp.contiguous()
        out = torch.empty(1, dtype=torch.float32, device=aux_logp.device)
        daux = torch.empty_like(aux_logp)
        _lib.check(lib.gscan_position_nll(aux_logp.data_ptr(), positions.data_ptr(), B, M, out.data_ptr(),
                                          daux.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "gscan_position_nll")
        ctx.save_for_backward(daux)
        ctx.batch = B
        return out[0] / B

    @staticmethod
    def backward(ctx, g):
        (daux,) = ctx.saved_tensors
        return daux * (g / ctx.batch), None


# ------------------------------------------------------------------------------------------
class Model(nn.Module):
    """Drop-in for `seq2seq.model.Model`.  Only the configuration the reference itself can run is
    supported: Bahdanau attention, simple situation representation, one decoder layer (up to four encoder layers)."""

    def __init__(self, input_vocabulary_size: int, embedding_dimension: int, encoder_hidden_size: int,
                 num_encoder_layers: int, target_vocabulary_size: int, encoder_dropout_p: float,
                 encoder_bidirectional: bool, num_decoder_layers: int, decoder_dropout_p: float,
                 decoder_hidden_size: int, num_cnn_channels: int, cnn_kernel_size: int,
                 cnn_dropout_p: float, cnn_hidden_num_channels: int, input_padding_idx: int, target_pad_idx: int,
                 target_eos_idx: int, output_directory: str, conditional_attention: bool, auxiliary_task: bool,
                 simple_situation_representation: bool, attention_type: str, **kwargs):
        super().__init__()
        if attention_type not in ("bahdanau", "luong"):
            raise ValueError("Unknown attention type {} specified.".format(attention_type))  # model.py:95-96
        if attention_type != "bahdanau":
            raise NotImplementedError("only Bahdanau attention is functional in the reference (SURVEY.md App. B)")
        if not simple_situation_representation:
            raise NotImplementedError("image situation representation is rejected by the reference CLI "
                                      "(__main__.py:112-114)")
        if num_decoder_layers != 1:
            raise NotImplementedError("one decoder layer is supported (more cannot work in the reference either, "
                                      "SURVEY.md App. B)")
        if not 1 <= num_encoder_layers <= _lib.MAX_ENC_LAYERS:
            raise NotImplementedError(f"1 to {_lib.MAX_ENC_LAYERS} encoder layers are supported")

        # construction order = RNG consumption order of the reference (model.py:47-87)
        self.situation_encoder = _WorldEncoderParams(num_cnn_channels, cnn_hidden_num_channels, cnn_kernel_size)
        self.visual_attention = _AttentionParams(cnn_hidden_num_channels * 3, decoder_hidden_size,
                                                 decoder_hidden_size)
        self.encoder = _CommandEncoderParams(input_vocabulary_size, embedding_dimension, encoder_hidden_size,
                                             num_encoder_layers, encoder_bidirectional, input_padding_idx)
        self.enc_hidden_to_dec_hidden = _HipLinear(encoder_hidden_size, decoder_hidden_size)
        self.tanh = nn.Tanh()
        self.textual_attention = _AttentionParams(encoder_hidden_size, decoder_hidden_size, decoder_hidden_size)
        self.attention_decoder = _DecoderParams(decoder_hidden_size, target_vocabulary_size, num_decoder_layers,
                                                target_pad_idx, self.textual_attention, self.visual_attention,
                                                conditional_attention)

        self.simple_situation_representation = simple_situation_representation
        self.attention_type = attention_type
        self.auxiliary_task = auxiliary_task
        self.conditional_attention = conditional_attention
        self.encoder_bidirectional = encoder_bidirectional
        self.target_eos_idx = target_eos_idx
        self.target_pad_idx = target_pad_idx
        self.input_padding_idx = input_padding_idx
        self.output_directory = output_directory
        self.dropout_p = (float(cnn_dropout_p), float(encoder_dropout_p), float(decoder_dropout_p))
        self.trained_iterations = 0
        self.best_iteration = 0
        self.best_exact_match = 0
        self.best_accuracy = 0
        self._hyper = dict(C=num_cnn_channels, Co=cnn_hidden_num_channels, K3=cnn_kernel_size,
                           E=embedding_dimension, He=encoder_hidden_size, H=decoder_hidden_size,
                           Vi=input_vocabulary_size, V=target_vocabulary_size, NL=int(num_encoder_layers))

        self._flat: Optional[torch.Tensor] = None
        self._flat_grad: Optional[torch.Tensor] = None
        self._param_struct = self._grad_struct = None
        self._workspace: Optional[torch.Tensor] = None
        self._generation = 0
        self._host_masks = None
        self._kernel_drop = None         # (seed, Philox stream id) of dropout the next launch draws inside its kernels
        self._dropout_seed = int(kwargs.get("seed", 42))
        self._dropout_calls = 0
        self._dropout_rank = 0
        self._mask_buffer = None
        self._predrawn = None            # (sizes, Philox stream id, device) of masks the optimiser launch drew ahead
        self._mask_buffer_deep = None
        self._dummy_aux = None
        self._mask_key = None
        self._mask_caps = None           # capacities of the three mask segments (grow-only, _draw_masks)
        self._anchor = None
        self._decode_state = None        # dims / batch of the last encode_input (greedy decoding)
        self._flatten()

    # ---- flat parameter / gradient buffers ------------------------------------------------
    def _named_unique(self) -> "OrderedDict[str, nn.Parameter]":
        return OrderedDict(self.named_parameters())

    def _flatten(self) -> None:
        """Re-home every parameter as a view of one contiguous fp32 buffer (same for .grad)."""
        params = self._named_unique()
        some = next(iter(params.values()))
        device = some.device
        # Every parameter starts on a 16-byte boundary of the buffer (up to three zero floats of padding in front of it:
        # the embedding tables have odd sizes and would leave every matrix behind them on a 4-byte boundary, where the
        # GEMM kernels fall back to one-float loads).  The padding takes part in the optimiser and the all-reduce as
        # zeros with zero gradients and stays zero.
        align = 4
        total = sum((p.numel() + align - 1) // align * align for p in params.values())
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        # four extra floats behind the gradients: the loss statistics of a data-parallel step travel in the same
        # all-reduce as the gradients (train.TrainStep)
        self._grad_store = torch.zeros(total + 4, dtype=torch.float32, device=device)
        grad = self._grad_store[:total]
        self._offsets: Dict[str, Tuple[int, int]] = {}
        off = 0
        for name, p in params.items():
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1).to(torch.float32))
            p.data = flat[off:off + n].view(p.shape)
            p.grad = None
            self._offsets[name] = (off, n)
            off += (n + align - 1) // align * align
        self._flat, self._flat_grad = flat, grad
        self._workspace = None
        self._anchor = torch.zeros((), dtype=torch.float32, device=device, requires_grad=True)
        self._param_struct = self._make_struct(flat)
        self._grad_struct = self._make_struct(grad)

    def _make_struct(self, flat: torch.Tensor) -> _lib.Params:
        s = _lib.Params()
        base = flat.data_ptr()
        for field, name in _lib.PARAM_FIELDS:
            if name in self._offsets:
                setattr(s, field, base + 4 * self._offsets[name][0])
            else:
                setattr(s, field, None)
        for layer in range(1, self._hyper["NL"]):
            for slot, pattern in enumerate(_lib.ENC_DEEP_FIELDS):
                name = "encoder.lstm." + pattern.format(layer)
                s.enc_deep[layer - 1][slot] = (base + 4 * self._offsets[name][0]) if name in self._offsets else None
        return s

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._flatten()
        return out

    @property
    def flat_parameters(self) -> torch.Tensor:
        """All parameters as one fp32 vector (named_parameters() order, each on a 16-byte boundary: `_offsets`)."""
        return self._flat

    @property
    def parameter_count(self) -> int:
        """Number of parameters (the flat buffer is a few floats longer: alignment padding)."""
        return sum(n for _, n in self._offsets.values())

    @property
    def flat_gradients(self) -> torch.Tensor:
        return self._flat_grad

    def attach_gradients(self, zero: bool) -> None:
        """Make every `param.grad` a view of the flat gradient buffer.  A parameter whose grad was
        dropped (zero_grad(set_to_none=True), the torch default) has its slice zeroed first."""
        if zero:
            self._flat_grad.zero_()
        for name, p in self._named_unique().items():
            off, n = self._offsets[name]
            view = self._flat_grad[off:off + n].view(p.shape)
            if p.grad is None:
                if not zero:
                    view.zero_()
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view

    # ---- dropout ----------------------------------------------------------------------------
    def set_dropout_masks(self, cnn: Optional[torch.Tensor], enc: Optional[torch.Tensor],
                          dec: Optional[torch.Tensor], enc_deep: Optional[torch.Tensor] = None) -> None:
        """Host-mask parity mode: use these scaled keep-masks ([B,G*G,3Co], [B,L,E], [B,T,H] and, with more than
        one encoder layer, [layers-1,B,L,D*He] for the inputs of layers 1..) for the next forward call instead of
        drawing them on the device (SURVEY.md §7 hard part 3)."""
        self._host_masks = (cnn, enc, dec, enc_deep)

    def set_dropout_rank(self, rank: int) -> None:
        """Data parallelism: rank r draws from its own Philox streams, so row i of every shard does not get the mask
        row i of every other shard gets (streams keyed by (seed, rank, step))."""
        if not 0 <= int(rank) < (1 << 16):
            raise ValueError(f"dropout rank {rank} out of range")
        self._dropout_rank = int(rank)

    def _philox_stream(self, deep: bool = False) -> int:
        """64-bit Philox stream id of the next mask draw: [rank (16 bits) | 0 = step masks, 1 = inter-layer masks
        (4 bits) | draw counter (40 bits)]; the seed is the Philox key."""
        return (self._dropout_rank << 44) | (int(deep) << 40) | (self._dropout_calls & ((1 << 40) - 1))

    def _deep_masks(self, B: int, L: int, device, deep_stream: int) -> Optional[torch.Tensor]:
        """nn.LSTM(dropout=p) drops the outputs of every layer but the last (seq2seq_model.py:44-45): masks in memory."""
        h = self._hyper
        if h["NL"] <= 1 or self.dropout_p[1] <= 0.0:
            return None
        D = 2 if self.encoder_bidirectional else 1
        shape = (h["NL"] - 1, B, L, D * h["He"])
        n = shape[0] * shape[1] * shape[2] * shape[3]
        if self._mask_buffer_deep is None or self._mask_buffer_deep.numel() != n:
            self._mask_buffer_deep = torch.empty(n, dtype=torch.float32, device=device)
        _lib.check(_lib.load().gscan_dropout_mask(self._mask_buffer_deep.data_ptr(), n, self.dropout_p[1],
                                                  self._dropout_seed, deep_stream,
                                                  torch.cuda.current_stream().cuda_stream), "gscan_dropout_mask")
        return self._mask_buffer_deep.view(shape)

    def _draw_masks(self, B: int, L: int, T: int, M: int, device, materialize: bool = False) -> Tuple[Optional[torch.Tensor], ...]:
        """The dropout of the next training-mode forward call.  Production mode (SURVEY.md 7 hard part 3, round 5): the
        masks are DRAWN INSIDE THE KERNELS that apply them (csrc/dropout.h) — this only fixes (seed, Philox stream id)
        for `_launch_forward`, returns no tensors and launches nothing.  `materialize=True` returns the same masks as
        tensors instead (gscan_dropout_masks_kernel_layout; a step fed with them through `set_dropout_masks` equals the
        in-kernel step bit for bit).  GSCAN_DROPOUT_IN_KERNEL=0: round 4's masks in memory, drawn by one Philox launch
        (or by the previous step's optimiser launch)."""
        self._kernel_drop = None
        if self._host_masks is not None:
            masks, self._host_masks = self._host_masks, None
            return tuple(None if m is None else m.to(device=device, dtype=torch.float32).contiguous()
                         for m in masks)
        if not self.training or max(self.dropout_p) <= 0.0:
            return (None, None, None)
        if _dropout_in_kernel():
            stream_id, deep_stream = self._philox_stream(), self._philox_stream(deep=True)
            self._dropout_calls += 1
            deep = self._deep_masks(B, L, device, deep_stream)
            if materialize:
                h = self._hyper
                out = [torch.empty(s, dtype=torch.float32, device=device)
                       for s in ((B, M, 3 * h["Co"]), (B, L, h["E"]), (B, T, h["H"]))]
                dims = self._dims(B, L, T, int(round(M ** 0.5)))
                _lib.check(_lib.load().gscan_dropout_masks_kernel_layout(
                    C.byref(dims), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), self.dropout_p[0],
                    self.dropout_p[1], self.dropout_p[2], self._dropout_seed, stream_id,
                    torch.cuda.current_stream().cuda_stream), "gscan_dropout_masks_kernel_layout")
                out = [m if p > 0.0 else None for m, p in zip(out, self.dropout_p)]
                return tuple(out) + ((deep,) if deep is not None else ())
            self._kernel_drop = (self._dropout_seed, stream_id)
            return (None, None, None) + ((deep,) if deep is not None else ())
        # the three masks of a step are one buffer filled by one Philox launch (cnn | enc | dec)
        lib = _lib.load()
        h = self._hyper
        shapes = ((B, M, 3 * h["Co"]), (B, L, h["E"]), (B, T, h["H"]))
        sizes = [s[0] * s[1] * s[2] for s in shapes]
        # One buffer of three segments (cnn | enc | dec), each with a CAPACITY that only grows: the largest mask of its kind
        # seen so far.  A draw always fills the capacities, and a batch takes the first n elements of each segment — the
        # elements of a counter-based mask are independent, any prefix of a segment is a mask.  So the masks that the
        # optimiser launch of the previous step drew ahead (TrainStep, gscan_adam_step_masks) serve the next batch
        # WHATEVER its padded lengths are: a file-fed loop, whose batches change shape from step to step (length buckets:
        # every step), launches nothing here.  (Until round 4 they were keyed by the exact shape: a batch of another
        # shape drew again, and the masks drawn ahead were wasted.)
        caps = self._mask_caps
        if caps is None or any(n > c for n, c in zip(sizes, caps)) or self._mask_buffer is None \
                or self._mask_buffer.device != device:
            caps = [max(n, c) for n, c in zip(sizes, caps or (0, 0, 0))]
            self._mask_caps = caps
            self._mask_buffer = torch.empty(sum(caps), dtype=torch.float32, device=device)
            self._predrawn = None
        self._mask_key = tuple(caps)
        buf = self._mask_buffer
        # drawn already for this position of the Philox counter?  Then there is nothing to launch.
        if self._predrawn == (tuple(caps), self._philox_stream(), device):
            self._predrawn = None
        else:
            self._predrawn = None
            _lib.check(lib.gscan_dropout_masks(buf.data_ptr(), caps[0], caps[1], caps[2], self.dropout_p[0],
                                               self.dropout_p[1], self.dropout_p[2], self._dropout_seed,
                                               self._philox_stream(), None,
                                               torch.cuda.current_stream().cuda_stream), "gscan_dropout_masks")
        deep_stream = self._philox_stream(deep=True)
        self._dropout_calls += 1
        out, off = [], 0
        for shape, n, cap, p in zip(shapes, sizes, caps, self.dropout_p):
            out.append(buf[off:off + n].view(shape) if p > 0.0 else None)
            off += cap
        deep = self._deep_masks(B, L, device, deep_stream)
        if deep is not None:
            out.append(deep)
        return tuple(out)

    # ---- the two launches ---------------------------------------------------------------------
    def _dims(self, B: int, L: int, T: int, G: int) -> _lib.Dims:
        h = self._hyper
        return _lib.Dims(B=B, L=L, T=T, G=G, C=h["C"], Co=h["Co"], K3=h["K3"], E=h["E"], He=h["He"], H=h["H"],
                         Vi=h["Vi"], V=h["V"], conditional=int(self.conditional_attention),
                         auxiliary=int(self.auxiliary_task), bidirectional=int(self.encoder_bidirectional),
                         pad_in=self.input_padding_idx, pad_tgt=self.target_pad_idx, enc_layers=h["NL"])

    def _require_device(self, *tensors) -> None:
        if not (self._flat.is_cuda and all(t.is_cuda for t in tensors)):
            raise RuntimeError("Model.forward runs on the HIP device only: move the model and the batch to "
                               "'cuda' (there is no CPU fallback in this package)")

    def _launch_forward(self, commands, lengths, world, targets, masks, positions=None, train_nll=None):
        lib = _lib.load()
        self._require_device(commands, world, targets)
        B, L = commands.shape
        _, T = targets.shape
        G = world.shape[1]
        if world.shape != (B, G, G, self._hyper["C"]):
            raise ValueError(f"situations_input must be [B,G,G,{self._hyper['C']}], got {tuple(world.shape)}")
        assert lengths.numel() == B, "Wrong amount of lengths passed to .forward()"   # seq2seq_model.py:57
        dims = self._dims(B, L, T, G)
        need = lib.gscan_workspace_bytes(C.byref(dims))
        if need == 0:
            raise _lib.GscanError("unsupported dimensions: " + lib.gscan_last_error().decode())
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=commands.device)
        commands = commands.contiguous()
        targets = targets.contiguous()
        world = _world_operand(world)
        if positions is not None:
            positions = positions.view(-1).contiguous()
        batch = _lib.Batch(commands.data_ptr(), lengths.data_ptr(), None if world.dtype == torch.uint8 else world.data_ptr(),
                           targets.data_ptr(), _lib.ptr(positions), world.data_ptr() if world.dtype == torch.uint8 else None)
        mstruct = _lib.Masks(*[_lib.ptr(m) for m in masks])
        if self._kernel_drop is not None:      # dropout drawn inside the kernels: (seed, stream id) fixed by _draw_masks
            mstruct.in_kernel = 1
            mstruct.p_cnn, mstruct.p_enc, mstruct.p_dec = self.dropout_p
            mstruct.seed, mstruct.stream_id = self._kernel_drop
            self._kernel_drop = None
        logp = torch.empty(B, T, self._hyper["V"], dtype=torch.float32, device=commands.device)
        aux = torch.empty(B, G * G, dtype=torch.float32, device=commands.device) if self.auxiliary_task else None
        if train_nll is None:
            _lib.check(lib.gscan_forward(C.byref(dims), C.byref(self._param_struct), C.byref(batch), C.byref(mstruct),
                                         self._workspace.data_ptr(), logp.data_ptr(), _lib.ptr(aux),
                                         torch.cuda.current_stream().cuda_stream), "gscan_forward")
        else:       # forward + training loss + backward in one call (train.TrainStep)
            weight_target_loss, sum_reduction, stats, seeds = train_nll
            _lib.check(lib.gscan_train_step_nll(C.byref(dims), C.byref(self._param_struct), C.byref(batch), C.byref(mstruct),
                                                self._workspace.data_ptr(), logp.data_ptr(), _lib.ptr(aux),
                                                float(weight_target_loss), int(sum_reduction), stats.data_ptr(),
                                                seeds.data_ptr(), C.byref(self._grad_struct),
                                                torch.cuda.current_stream().cuda_stream), "gscan_train_step_nll")
        self._generation += 1
        call = dict(dims=dims, batch=batch, masks=mstruct, generation=self._generation,
                    keep=(commands, lengths, world, targets, masks, positions))
        if aux is None:
            if self._dummy_aux is None or self._dummy_aux.device != commands.device:
                self._dummy_aux = torch.zeros(1, device=commands.device)
            aux = self._dummy_aux
        return logp, aux, call

    def _launch_backward(self, call, dlogp: torch.Tensor, daux: Optional[torch.Tensor],
                         seeds: Optional[torch.Tensor] = None, attach: bool = True) -> None:
        lib = _lib.load()
        if call["generation"] != self._generation:
            raise RuntimeError("backward() after another forward(): the saved activations were overwritten "
                               "(one in-flight step per model, as in the reference's training loop)")
        if attach:
            self.attach_gradients(zero=False)
        dlogp = dlogp.contiguous()
        daux = None if daux is None else daux.contiguous()
        _lib.check(lib.gscan_backward_seeded(C.byref(call["dims"]), C.byref(self._param_struct),
                                             C.byref(call["batch"]), C.byref(call["masks"]),
                                             self._workspace.data_ptr(), dlogp.data_ptr(), _lib.ptr(daux),
                                             _lib.ptr(seeds), C.byref(self._grad_struct),
                                             torch.cuda.current_stream().cuda_stream), "gscan_backward_seeded")

    def _launch_backward_nll(self, call, weight_target_loss: float, stats: torch.Tensor, seeds: torch.Tensor,
                             sum_reduction: bool = False) -> None:
        """loss.backward() of the training loss itself (train.py:102-110): seeded inside the backward kernels from
        the per-row loss partials the forward pass left in the workspace; fills stats[4] and seeds[3]."""
        lib = _lib.load()
        if call["generation"] != self._generation:
            raise RuntimeError("backward() after another forward(): the saved activations were overwritten "
                               "(one in-flight step per model, as in the reference's training loop)")
        _lib.check(lib.gscan_backward_nll(C.byref(call["dims"]), C.byref(self._param_struct), C.byref(call["batch"]),
                                          C.byref(call["masks"]), self._workspace.data_ptr(),
                                          float(weight_target_loss), int(sum_reduction), stats.data_ptr(),
                                          seeds.data_ptr(),
                                          C.byref(self._grad_struct), torch.cuda.current_stream().cuda_stream),
                   "gscan_backward_nll")

    def workspace_view(self, call_dims: _lib.Dims, name: str) -> torch.Tensor:
        """A saved activation of the last forward/backward as a flat fp32 tensor (tests, debugging)."""
        lib = _lib.load()
        off, cnt = C.c_size_t(), C.c_size_t()
        _lib.check(lib.gscan_workspace_find(C.byref(call_dims), name.encode(), C.byref(off), C.byref(cnt)),
                   "gscan_workspace_find")
        return self._workspace[off.value:off.value + 4 * cnt.value].view(torch.float32)

    # ---- the reference's surface ------------------------------------------------------------
    def forward(self, commands_input: torch.LongTensor, commands_lengths: List[int],
                situations_input: torch.Tensor, target_batch: torch.LongTensor,
                target_lengths: List[int]) -> Tuple[torch.Tensor, torch.Tensor]:
        """model.py:206-219.  Returns (log-probabilities [B,T,V], auxiliary log-scores [B,G*G]); without
        the auxiliary task the second element is the reference's dummy pair of zeros(1)."""
        self._require_device(commands_input, situations_input, target_batch)
        device = commands_input.device
        lengths = _as_int32_lengths(commands_lengths, device)
        B, L = commands_input.shape
        masks = self._draw_masks(B, L, target_batch.shape[1], situations_input.shape[1] ** 2, device)
        if torch.is_grad_enabled():
            logp, aux = _NetworkFunction.apply(self._anchor, self, commands_input, lengths, situations_input,
                                               target_batch, masks)
        else:
            logp, aux, _ = self._launch_forward(commands_input, lengths, situations_input, target_batch, masks)
        if not self.auxiliary_task:
            return logp, (torch.zeros(1), torch.zeros(1))      # model.py:217
        return logp, aux

    def get_loss(self, target_scores: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        return _SequenceNLL.apply(target_scores, targets.contiguous(), self.target_pad_idx)

    def get_auxiliary_loss(self, auxiliary_scores_target: torch.Tensor, target_target_positions: torch.Tensor):
        return _PositionNLL.apply(auxiliary_scores_target, target_target_positions.view(-1).contiguous())

    def auxiliary_task_forward(self, output_scores_target_pos: torch.Tensor) -> torch.Tensor:
        assert self.auxiliary_task, "Please set auxiliary_task to True if using it."
        return torch.log_softmax(output_scores_target_pos, -1)

    def get_metrics(self, target_scores: torch.Tensor, targets: torch.Tensor) -> Tuple[float, float]:
        """model.py:117-137: (token accuracy %, exact match %) under the padding mask."""
        lib = _lib.load()
        B, T, V = target_scores.shape
        out = torch.empty(3, dtype=torch.float32, device=target_scores.device)
        scores = target_scores.detach().contiguous()
        _lib.check(lib.gscan_sequence_metrics(scores.data_ptr(), targets.contiguous().data_ptr(), B, T, V,
                                              self.target_pad_idx, out.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "gscan_sequence_metrics")
        correct, live, exact = out.tolist()
        return 100.0 * correct / live, 100.0 * exact / B

    @staticmethod
    def get_auxiliary_accuracy(target_scores: torch.Tensor, targets: torch.Tensor) -> float:
        with torch.no_grad():
            hits = (target_scores.argmax(dim=1) == targets.view(-1)).sum().item()
        return 100.0 * hits / len(targets)

    @staticmethod
    def remove_start_of_sequence(input_tensor: torch.Tensor) -> torch.Tensor:
        """model.py:108-115: drop the SOS column, append a column of zeros."""
        return torch.cat([input_tensor[:, 1:], input_tensor.new_zeros(input_tensor.shape[0], 1)], dim=1)

    def update_state(self, is_best: bool, accuracy=None, exact_match=None) -> None:
        self.trained_iterations += 1
        if is_best:
            self.best_exact_match, self.best_accuracy = exact_match, accuracy
            self.best_iteration = self.trained_iterations

    # ---- greedy decoding surface (model.py:172-188, used by predict.py:82-106) ---------------------
    def encode_input(self, commands_input: torch.LongTensor, commands_lengths: List[int],
                     situations_input: torch.Tensor) -> Dict[str, torch.Tensor]:
        """model.py:172-180.  Runs both encoders — and, for the decode_input calls that follow, the key projections
        and the bridge — in one library call; returns the reference's dictionary."""
        lib = _lib.load()
        self._require_device(commands_input, situations_input)
        device = commands_input.device
        lengths = _as_int32_lengths(commands_lengths, device)
        B, L = commands_input.shape
        G = situations_input.shape[1]
        if situations_input.shape != (B, G, G, self._hyper["C"]):
            raise ValueError(f"situations_input must be [B,G,G,{self._hyper['C']}], got {tuple(situations_input.shape)}")
        assert lengths.numel() == B, "Wrong amount of lengths passed to .forward()"   # seq2seq_model.py:57
        dims = self._dims(B, L, 1, G)
        need = lib.gscan_workspace_bytes(C.byref(dims))
        if need == 0:
            raise _lib.GscanError("unsupported dimensions: " + lib.gscan_last_error().decode())
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=device)
        commands = commands_input.contiguous()
        world = _world_operand(situations_input)
        batch = _lib.Batch(commands.data_ptr(), lengths.data_ptr(), None if world.dtype == torch.uint8 else world.data_ptr(),
                           None, None, world.data_ptr() if world.dtype == torch.uint8 else None)
        masks = _lib.Masks(None, None, None)      # eval semantics: predict() calls model.eval() first (predict.py:70)
        _lib.check(lib.gscan_encode(C.byref(dims), C.byref(self._param_struct), C.byref(batch), C.byref(masks),
                                    self._workspace.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "gscan_encode")
        self._generation += 1
        M, F, He = G * G, 3 * self._hyper["Co"], self._hyper["He"]
        self._decode_state = dict(dims=dims, batch=batch, generation=self._generation, B=B, L=L, M=M,
                                  keep=(commands, lengths, world))
        feat = self.workspace_view(dims, "feat").view(B, M, F).clone()
        enc_out = self.workspace_view(dims, "enc_out").view(B, L, He).transpose(0, 1).contiguous()
        hidden = self.workspace_view(dims, "hN").view(B, He).clone()
        return {"encoded_situations": feat,
                "encoded_commands": {"encoder_outputs": enc_out, "sequence_lengths": commands_lengths},
                "hidden_states": hidden}

    def decode_input(self, target_token: torch.LongTensor, hidden: Tuple[torch.Tensor, torch.Tensor],
                     encoder_outputs: torch.Tensor, input_lengths: List[int],
                     encoded_situations: torch.Tensor):
        """model.py:182-188 = one decoder step (seq2seq_model.py:359-431) for all rows.  As in predict.py:87-106,
        `encoder_outputs` / `encoded_situations` are the PROJECTED keys of the latest encode_input; the step runs on
        the copies that call left on the device (the arguments are checked for shape only).  Returns the
        reference's tuple (logits, (h, c), alpha_vis, alpha_text, alpha_vis)."""
        lib = _lib.load()
        st = self._decode_state
        if st is None or st["generation"] != self._generation:
            raise RuntimeError("decode_input() needs the encode_input() of the same examples first "
                               "(the encoded memories live in the model's device workspace)")
        B, L, M, H = st["B"], st["L"], st["M"], self._hyper["H"]
        h, c = hidden
        if tuple(encoder_outputs.shape) != (L, B, H) or tuple(encoded_situations.shape) != (B, M, H):
            raise ValueError(f"projected keys must be [{L},{B},{H}] and [{B},{M},{H}], got "
                             f"{tuple(encoder_outputs.shape)} and {tuple(encoded_situations.shape)}")
        assert len(input_lengths) == B                                   # seq2seq_model.py:120
        device = h.device
        self._require_device(h, c, target_token)
        tokens = target_token.reshape(B).to(torch.int64).contiguous()
        h_in = h.reshape(B, H).to(torch.float32).contiguous()
        c_in = c.reshape(B, H).to(torch.float32).contiguous()
        V = self._hyper["V"]
        logits = torch.empty(B, V, dtype=torch.float32, device=device)
        h_out, c_out = torch.empty_like(h_in), torch.empty_like(c_in)
        alpha_text = torch.empty(B, L, dtype=torch.float32, device=device)
        alpha_vis = torch.empty(B, M, dtype=torch.float32, device=device)
        _lib.check(lib.gscan_decode_step(C.byref(st["dims"]), C.byref(self._param_struct), C.byref(st["batch"]),
                                         tokens.data_ptr(), h_in.data_ptr(), c_in.data_ptr(),
                                         self._workspace.data_ptr(), logits.data_ptr(), h_out.data_ptr(),
                                         c_out.data_ptr(), alpha_text.data_ptr(), alpha_vis.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "gscan_decode_step")
        return logits, (h_out.unsqueeze(0), c_out.unsqueeze(0)), alpha_vis, alpha_text, alpha_vis

    def greedy_decode(self, commands_input: torch.LongTensor, commands_lengths: List[int],
                      situations_input: torch.Tensor, sos_idx: int, eos_idx: int, max_decoding_steps: int):
        """predict.py:82-115 for every row of a batch in ONE library call (`gscan_greedy_decode`): encode, then the
        persistent decoder kernel feeds its own argmax back, each row until its <EOS> or max_decoding_steps + 1
        steps — no launch and no host synchronisation per token.  Returns device tensors (tokens [B,S] int64, steps
        [B] int32, alpha_text [B,S,L], alpha_vis [B,S,G*G], att_sum [B,G*G]), S = max_decoding_steps + 1; entries of
        a row behind its own `steps` are undefined."""
        lib = _lib.load()
        self._require_device(commands_input, situations_input)
        device = commands_input.device
        lengths = _as_int32_lengths(commands_lengths, device)
        B, L = commands_input.shape
        G = situations_input.shape[1]
        if situations_input.shape != (B, G, G, self._hyper["C"]):
            raise ValueError(f"situations_input must be [B,G,G,{self._hyper['C']}], got {tuple(situations_input.shape)}")
        assert lengths.numel() == B, "Wrong amount of lengths passed to .forward()"   # seq2seq_model.py:57
        dims = self._dims(B, L, 1, G)
        need = lib.gscan_workspace_bytes(C.byref(dims))
        if need == 0:
            raise _lib.GscanError("unsupported dimensions: " + lib.gscan_last_error().decode())
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=device)
        commands = commands_input.contiguous()
        world = _world_operand(situations_input)
        batch = _lib.Batch(commands.data_ptr(), lengths.data_ptr(), None if world.dtype == torch.uint8 else world.data_ptr(),
                           None, None, world.data_ptr() if world.dtype == torch.uint8 else None)
        S, M = int(max_decoding_steps) + 1, G * G
        tokens = torch.zeros(B, S, dtype=torch.int64, device=device)
        steps = torch.zeros(B, dtype=torch.int32, device=device)
        alpha_text = torch.zeros(B, S, L, dtype=torch.float32, device=device)
        alpha_vis = torch.zeros(B, S, M, dtype=torch.float32, device=device)
        att_sum = torch.zeros(B, M, dtype=torch.float32, device=device)
        _lib.check(lib.gscan_greedy_decode(C.byref(dims), S, C.byref(self._param_struct), C.byref(batch),
                                           self._workspace.data_ptr(), int(sos_idx), int(eos_idx), tokens.data_ptr(),
                                           steps.data_ptr(), alpha_text.data_ptr(), alpha_vis.data_ptr(),
                                           att_sum.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "gscan_greedy_decode")
        self._generation += 1
        self._decode_state = None
        return tokens, steps, alpha_text, alpha_vis, att_sum

    def decode_input_batched(self, target_batch: torch.LongTensor, target_lengths: List[int],
                             initial_hidden: torch.Tensor, encoded_commands: torch.Tensor,
                             command_lengths: List[int], encoded_situations: torch.Tensor):
        """model.py:190-204, the second half of the reference's forward(): teacher-forced decoding from encodings
        handed in (the dictionary entries of encode_input: hidden_states [B,He], encoder_outputs [L,B,He],
        encoded_situations [B,G*G,3Co]).  Returns the reference's pair (log-probabilities [T,B,V] TIME-major,
        summed visual attention [B,G*G]).  One library call (`gscan_decode_batched`); eval semantics, no autograd —
        training goes through forward(), which fuses both halves."""
        lib = _lib.load()
        self._require_device(target_batch, initial_hidden, encoded_commands, encoded_situations)
        device = target_batch.device
        B, T = target_batch.shape
        L, M = encoded_commands.shape[0], encoded_situations.shape[1]
        G = int(round(M ** 0.5))
        h = self._hyper
        if (tuple(encoded_commands.shape) != (L, B, h["He"]) or tuple(initial_hidden.shape) != (B, h["He"]) or
                tuple(encoded_situations.shape) != (B, G * G, 3 * h["Co"])):
            raise ValueError("decode_input_batched: expected hidden [B,He], encoder_outputs [L,B,He] and "
                             f"encoded_situations [B,G*G,{3 * h['Co']}], got {tuple(initial_hidden.shape)}, "
                             f"{tuple(encoded_commands.shape)}, {tuple(encoded_situations.shape)}")
        lengths = _as_int32_lengths(command_lengths, device)
        assert lengths.numel() == B                                       # seq2seq_model.py:120
        dims = self._dims(B, L, T, G)
        need = lib.gscan_workspace_bytes(C.byref(dims))
        if need == 0:
            raise _lib.GscanError("unsupported dimensions: " + lib.gscan_last_error().decode())
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = torch.empty(need, dtype=torch.uint8, device=device)
        targets = target_batch.contiguous()
        enc_out = encoded_commands.detach().to(torch.float32).transpose(0, 1).contiguous()
        feat = encoded_situations.detach().to(torch.float32).contiguous()
        hN = initial_hidden.detach().to(torch.float32).contiguous()
        batch = _lib.Batch(None, lengths.data_ptr(), None, targets.data_ptr(), None, None)
        logp = torch.empty(B, T, h["V"], dtype=torch.float32, device=device)
        att_sum = torch.empty(B, G * G, dtype=torch.float32, device=device)
        _lib.check(lib.gscan_decode_batched(C.byref(dims), C.byref(self._param_struct), C.byref(batch), feat.data_ptr(),
                                            enc_out.data_ptr(), hN.data_ptr(), self._workspace.data_ptr(),
                                            logp.data_ptr(), att_sum.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream), "gscan_decode_batched")
        self._generation += 1
        self._decode_state = None
        return logp.transpose(0, 1), att_sum

    # ---- checkpoints (model.py:228-261): same dictionary keys, same file names ----------------
    def get_current_state(self) -> dict:
        return {"iteration": self.trained_iterations, "state_dict": self.state_dict(),
                "best_iteration": self.best_iteration, "best_accuracy": self.best_accuracy,
                "best_exact_match": self.best_exact_match}

    def load_model(self, path_to_checkpoint: str) -> dict:
        checkpoint = torch.load(path_to_checkpoint, map_location=self._flat.device)
        self.load_state_dict(checkpoint["state_dict"])
        for key, attr in (("iteration", "trained_iterations"), ("best_iteration", "best_iteration"),
                          ("best_exact_match", "best_exact_match"), ("best_accuracy", "best_accuracy")):
            setattr(self, attr, checkpoint[key])
        return checkpoint["optimizer_state_dict"]

    def save_checkpoint(self, file_name: str, is_best: bool, optimizer_state_dict: dict) -> str:
        path = os.path.join(self.output_directory, file_name)
        state = self.get_current_state()
        state["optimizer_state_dict"] = optimizer_state_dict
        torch.save(state, path)
        if is_best:
            shutil.copyfile(path, os.path.join(self.output_directory, "model_best.pth.tar"))
        return path
