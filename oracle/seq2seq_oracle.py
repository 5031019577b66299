"""CPU oracle for the gSCAN seq2seq training step.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32 or fp64) *restatement* of the algorithm the
reference runs on its training hot path.  It exists so that the HIP kernels can be
checked against something that runs everywhere.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it; the product package ``multimodal_seq2seq_gscan_amd`` never does and fails loudly
when its HIP library is missing.

Pinning: the reference has no functional tests for this path, so the
oracle is pinned against outputs of the reference itself, generated in the authoring
container by ``tests/golden/make_golden.py`` (which imports ``/root/reference``) and
committed as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks it.

Every function cites the reference lines it restates (paths relative to the
reference checkout).  Parameters are addressed by the reference's ``state_dict``
names so a reference checkpoint can be fed in directly.

Conventions: B batch, L command length, T target length, G grid side, C grid
channels, F = 3*Cout conv features, He/H encoder/decoder hidden, E embedding.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

Params = Dict[str, torch.Tensor]


def _lengths(x, device) -> torch.Tensor:
    """Lengths arrive as python lists, numpy float64 arrays or tensors
    (seq2seq/gSCAN_dataset.py:276 builds them with np.append on a float array)."""
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.long)
    return torch.as_tensor([int(v) for v in x], dtype=torch.long, device=device)


# ----------------------------------------------------------------------------------
# a1  ConvolutionalNet.forward            seq2seq/cnn_model.py:22-36
# ----------------------------------------------------------------------------------
def world_encoder(p: Params, world: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """world [B,G,G,C] -> features [B,G*G,3*Cout].

    The reference transposes dims 1 and 3 before and after the convolutions
    (cnn_model.py:28,34), so with the stored weight W[o,ch,kh,kw] the tap (kh,kw)
    moves along (grid column, grid row) of the *untransposed* input.  Written here as
    an explicit gather of the k*k shifted views of the [B,row,col,C] tensor (tap-major
    im2col) and ONE product per convolution with the weights reordered to
    [tap, ch, o] — no conv2d call.  (Until round 6 this was a Python loop of k*k small
    products and adds: the same sum, 10 % of the oracle's CPU time at the benchmark
    shape, which bench.py's cpu_baseline times.)
    """
    B, G, _, C = world.shape
    outs = []
    for name in ("conv_1", "conv_2", "conv_3"):
        W = p[f"situation_encoder.{name}.weight"]
        bias = p[f"situation_encoder.{name}.bias"]
        Co, _, k, _ = W.shape
        pad = k // 2
        xp = torch.zeros(B, G + 2 * pad, G + 2 * pad, C, dtype=world.dtype)
        xp[:, pad:pad + G, pad:pad + G, :] = world
        # kh pairs with the grid *column* offset, kw with the grid *row* offset
        taps = [xp[:, kw:kw + G, kh:kh + G, :] for kh in range(k) for kw in range(k)]      # k*k views [B,G,G,C]
        cols = torch.stack(taps, dim=3).reshape(B, G, G, k * k * C)                        # [.., (kh, kw, ch)]
        Wm = W.permute(2, 3, 1, 0).reshape(k * k * C, Co)                                  # [(kh, kw, ch), o]
        outs.append(cols @ Wm + bias.view(1, 1, 1, Co))
    feat = torch.relu(torch.cat(outs, dim=-1))                     # order conv_1|conv_2|conv_3
    if mask is not None:                                           # nn.Dropout as a given scaled mask
        feat = feat * mask.view_as(feat)
    return feat.reshape(B, G * G, -1)


# ----------------------------------------------------------------------------------
# LSTM cell, gate order (i, f, g, o) as torch.nn.LSTM uses it
# ----------------------------------------------------------------------------------
def lstm_cell(x_proj: torch.Tensor, h: torch.Tensor, c: torch.Tensor, w_hh: torch.Tensor,
              ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """x_proj already holds W_ih x + b_ih + b_hh.  Returns (h', c', activated gates)."""
    H = h.shape[-1]
    pre = x_proj + h @ w_hh.t()
    i = torch.sigmoid(pre[..., 0 * H:1 * H])
    f = torch.sigmoid(pre[..., 1 * H:2 * H])
    g = torch.tanh(pre[..., 2 * H:3 * H])
    o = torch.sigmoid(pre[..., 3 * H:4 * H])
    c2 = f * c + i * g
    h2 = o * torch.tanh(c2)
    return h2, c2, torch.cat([i, f, g, o], dim=-1)


# ----------------------------------------------------------------------------------
# a2  EncoderRNN.forward                  seq2seq/seq2seq_model.py:47-89
# ----------------------------------------------------------------------------------
def command_encoder(p: Params, tokens: torch.Tensor, lengths, bidirectional: bool = True,
                    mask: Optional[torch.Tensor] = None, layer_masks: Optional[Sequence[torch.Tensor]] = None,
                    ) -> Tuple[torch.Tensor, torch.Tensor]:
    """tokens [B,L] -> (final hidden [B,He], per-step outputs [B,L,He]).

    The reference sorts, packs, runs nn.LSTM and unpacks (:62-88).  Equivalent
    statement used here: each direction walks only the first len_b tokens of row b
    (the reverse direction starts at token len_b-1), outputs at padded positions
    are zero, and the two directions are summed (:77-80).  With more than one layer
    (nn.LSTM(num_layers=n), :44-45) the input of layer l+1 is the concatenation
    [forward | reverse] of layer l's outputs, dropped out in training
    (``layer_masks[l]``: scaled keep mask [B,L,D*He]); the direction sums and the
    final hidden state are taken from the LAST layer only (:76-82).
    """
    B, L = tokens.shape
    lens = _lengths(lengths, tokens.device)
    x = p["encoder.embedding.weight"][tokens]                      # padding row is zero at init
    if mask is not None:
        x = x * mask.view_as(x)
    He = p["encoder.lstm.weight_hh_l0"].shape[1]
    layers = 1
    while f"encoder.lstm.weight_hh_l{layers}" in p:
        layers += 1
    suffixes = [""] + (["_reverse"] if bidirectional else [])
    for layer in range(layers):
        per_dir, finals = [], []
        for suffix in suffixes:
            w_ih = p[f"encoder.lstm.weight_ih_l{layer}{suffix}"]
            w_hh = p[f"encoder.lstm.weight_hh_l{layer}{suffix}"]
            b = p[f"encoder.lstm.bias_ih_l{layer}{suffix}"] + p[f"encoder.lstm.bias_hh_l{layer}{suffix}"]
            xp = x @ w_ih.t() + b
            h = torch.zeros(B, He, dtype=x.dtype)
            c = torch.zeros(B, He, dtype=x.dtype)
            out = torch.zeros(B, L, He, dtype=x.dtype)
            steps = range(L - 1, -1, -1) if suffix else range(L)
            for t in steps:
                live = (t < lens).to(x.dtype).unsqueeze(1)
                h2, c2, _ = lstm_cell(xp[:, t], h, c, w_hh)
                h = live * h2 + (1 - live) * h
                c = live * c2 + (1 - live) * c
                out[:, t] = live * h2
            per_dir.append(out)
            finals.append(h)
        if layer + 1 < layers:
            x = torch.cat(per_dir, dim=2)
            if layer_masks is not None and layer_masks[layer] is not None:
                x = x * layer_masks[layer].view_as(x)
    return sum(finals), sum(per_dir)


# ----------------------------------------------------------------------------------
# a4  Attention.forward                   seq2seq/seq2seq_model.py:105-139
# ----------------------------------------------------------------------------------
def additive_attention(prefix: str, p: Params, query: torch.Tensor, proj_keys: torch.Tensor,
                       mem_lengths: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """query [B,Hq], proj_keys [B,M,H] (also used as the values, seq2seq_model.py:476-478)
    -> (context [B,H], weights [B,M])."""
    q = query @ p[f"{prefix}.query_layer.weight"].t()
    v = p[f"{prefix}.energy_layer.weight"].view(-1)
    scores = torch.tanh(q.unsqueeze(1) + proj_keys) @ v            # [B,M]
    M = proj_keys.shape[1]
    dead = torch.arange(M).unsqueeze(0) >= mem_lengths.unsqueeze(1)  # helpers.py:11-32
    scores = scores.masked_fill(dead, float("-inf"))
    w = torch.softmax(scores, dim=1)
    ctx = (w.unsqueeze(2) * proj_keys).sum(dim=1)
    return ctx, w


# ----------------------------------------------------------------------------------
# a3+a5+a6+a7  Model.decode_input_batched / BahdanauAttentionDecoderRNN.forward
#              seq2seq/model.py:190-219, seq2seq/seq2seq_model.py:359-492
# ----------------------------------------------------------------------------------
def decoder(p: Params, hN: torch.Tensor, enc_out: torch.Tensor, cmd_lengths, feats: torch.Tensor,
            targets: torch.Tensor, conditional: bool, mask: Optional[torch.Tensor] = None,
            keep: Optional[dict] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Teacher-forced decode of all T steps for all rows (no target-length masking,
    seq2seq_model.py:473).  Returns (logits [B,T,V], summed visual attention [B,G*G]).

    The reference's length sort/unsort (:454-463,485-488) is a pure permutation of
    independent rows and is omitted.
    """
    B, T = targets.shape
    lens = _lengths(cmd_lengths, targets.device)
    h = torch.tanh(hN @ p["enc_hidden_to_dec_hidden.weight"].t() + p["enc_hidden_to_dec_hidden.bias"])
    c = h.clone()                                                  # seq2seq_model.py:494-504
    pk_vis = feats @ p["visual_attention.key_layer.weight"].t()    # :466-467
    pk_txt = enc_out @ p["textual_attention.key_layer.weight"].t()  # :468-469
    emb_all = p["attention_decoder.embedding.weight"][targets]     # [B,T,H]
    if mask is not None:
        emb_all = emb_all * mask.view_as(emb_all)
    w_ih = p["attention_decoder.lstm.weight_ih_l0"]
    w_hh = p["attention_decoder.lstm.weight_hh_l0"]
    b = p["attention_decoder.lstm.bias_ih_l0"] + p["attention_decoder.lstm.bias_hh_l0"]
    w_o2h = p["attention_decoder.output_to_hidden.weight"]
    w_h2o = p["attention_decoder.hidden_to_output.weight"]
    full = torch.full((B,), pk_vis.shape[1], dtype=torch.long)
    logits = []
    att_sum = torch.zeros(B, pk_vis.shape[1], dtype=feats.dtype)
    for t in range(T):
        e = emb_all[:, t]
        ctx_c, a_c = additive_attention("textual_attention", p, h, pk_txt, lens)        # :388-390
        if conditional:                                                                   # :394-396
            q = torch.tanh(torch.cat([h, ctx_c], dim=1) @ p["attention_decoder.queries_to_keys.weight"].t()
                           + p["attention_decoder.queries_to_keys.bias"])
        else:
            q = h
        ctx_s, a_s = additive_attention("visual_attention", p, q, pk_vis, full)         # :400-402
        x = torch.cat([e, ctx_c, ctx_s], dim=1)
        h, c, gates = lstm_cell(x @ w_ih.t() + b, h, c, w_hh)                            # :414
        pre = torch.cat([e, h, ctx_c, ctx_s], dim=1) @ w_o2h.t()                         # :421-423
        logits.append(pre @ w_h2o.t())                                                    # :424
        att_sum = att_sum + a_s                                                           # :479,490
        if keep is not None:
            keep.setdefault("h", []).append(h)
            keep.setdefault("c", []).append(c)
            keep.setdefault("a_c", []).append(a_c)
            keep.setdefault("a_s", []).append(a_s)
            keep.setdefault("ctx_c", []).append(ctx_c)
            keep.setdefault("ctx_s", []).append(ctx_s)
            keep.setdefault("gates", []).append(gates)
    return torch.stack(logits, dim=1), att_sum


def greedy_decode(p: Params, commands: torch.Tensor, cmd_lengths, world: torch.Tensor, sos_idx: int, eos_idx: int,
                  max_decoding_steps: int, conditional: bool, bidirectional: bool = True) -> List[dict]:
    """seq2seq/predict.py:57-128 restated: for every row on its own, encode, then feed back the argmax token until
    EOS or until `decoding_iteration <= max_decoding_steps` fails (so at most max_decoding_steps + 1 steps).
    Returns per row: tokens (with the final EOS if one was produced), per-step logits and both attention rows."""
    feats = world_encoder(p, world)
    hN, enc_out = command_encoder(p, commands, cmd_lengths, bidirectional=bidirectional)
    lens = _lengths(cmd_lengths, commands.device)
    pk_vis_all = feats @ p["visual_attention.key_layer.weight"].t()            # predict.py:87-88
    pk_txt_all = enc_out @ p["textual_attention.key_layer.weight"].t()         # :89-90
    w_ih = p["attention_decoder.lstm.weight_ih_l0"]
    w_hh = p["attention_decoder.lstm.weight_hh_l0"]
    bias = p["attention_decoder.lstm.bias_ih_l0"] + p["attention_decoder.lstm.bias_hh_l0"]
    w_o2h = p["attention_decoder.output_to_hidden.weight"]
    w_h2o = p["attention_decoder.hidden_to_output.weight"]
    rows = []
    for r in range(commands.shape[0]):
        pk_vis, pk_txt, n = pk_vis_all[r:r + 1], pk_txt_all[r:r + 1], lens[r:r + 1]
        full = torch.full((1,), pk_vis.shape[1], dtype=torch.long)
        h = torch.tanh(hN[r:r + 1] @ p["enc_hidden_to_dec_hidden.weight"].t() + p["enc_hidden_to_dec_hidden.bias"])
        c = h.clone()                                                          # :95-96
        token, it = sos_idx, 0
        out = {"tokens": [], "logits": [], "alpha_text": [], "alpha_vis": []}
        while token != eos_idx and it <= max_decoding_steps:                   # :101
            e = p["attention_decoder.embedding.weight"][torch.tensor([token])]
            ctx_c, a_c = additive_attention("textual_attention", p, h, pk_txt, n)
            if conditional:
                q = torch.tanh(torch.cat([h, ctx_c], dim=1) @ p["attention_decoder.queries_to_keys.weight"].t()
                               + p["attention_decoder.queries_to_keys.bias"])
            else:
                q = h
            ctx_s, a_s = additive_attention("visual_attention", p, q, pk_vis, full)
            h, c, _ = lstm_cell(torch.cat([e, ctx_c, ctx_s], dim=1) @ w_ih.t() + bias, h, c, w_hh)
            logit = (torch.cat([e, h, ctx_c, ctx_s], dim=1) @ w_o2h.t()) @ w_h2o.t()
            token = int(torch.log_softmax(logit, dim=-1).max(dim=-1)[1].item())   # :106-107
            out["tokens"].append(token)
            out["logits"].append(logit[0])
            out["alpha_text"].append(a_c[0])
            out["alpha_vis"].append(a_s[0])
            it += 1
        rows.append(out)
    return rows


# ----------------------------------------------------------------------------------
# a7  Model.forward                       seq2seq/model.py:206-219
# ----------------------------------------------------------------------------------
def forward(p: Params, commands: torch.Tensor, cmd_lengths, world: torch.Tensor, targets: torch.Tensor,
            conditional: bool = True, auxiliary: bool = False, bidirectional: bool = True,
            masks: Optional[Sequence[Optional[torch.Tensor]]] = None, keep: Optional[dict] = None,
            ) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """Returns (log-probs [B,T,V], aux log-scores [B,G*G] or None).

    ``masks`` = (cnn, encoder-embedding, decoder-embedding[, inputs of encoder layers 1..]) scaled dropout masks
    (value 0 or 1/(1-p)) or None for no dropout.  Decoder mask is in batch row order.
    """
    m_cnn, m_enc, m_dec = tuple(masks[:3]) if masks is not None else (None, None, None)
    m_deep = masks[3] if masks is not None and len(masks) > 3 else None        # [layers-1, B, L, D*He]
    feats = world_encoder(p, world, m_cnn)
    hN, enc_out = command_encoder(p, commands, cmd_lengths, bidirectional, m_enc, m_deep)
    logits, att_sum = decoder(p, hN, enc_out, cmd_lengths, feats, targets, conditional, m_dec, keep)
    if keep is not None:
        keep.update(feats=feats, hN=hN, enc_out=enc_out, logits=logits, att_sum=att_sum)
    logp = torch.log_softmax(logits, dim=-1)                       # model.py:203
    aux = torch.log_softmax(att_sum, dim=-1) if auxiliary else None  # model.py:166-170
    return logp, aux


# ----------------------------------------------------------------------------------
# a8  Model.get_loss                      seq2seq/model.py:108-115,147-160
# ----------------------------------------------------------------------------------
def shift_targets(targets: torch.Tensor, pad_idx: int = 0) -> torch.Tensor:
    """Drop SOS, append one column.  The reference appends literal zeros (model.py:112),
    which equals pad for every vocabulary it builds (gSCAN_dataset.py:22-32)."""
    B = targets.shape[0]
    return torch.cat([targets[:, 1:], torch.zeros(B, 1, dtype=targets.dtype)], dim=1)


def sequence_loss(logp: torch.Tensor, targets: torch.Tensor, pad_idx: int = 0, reduction: str = "mean"):
    """NLL over positions whose shifted target != pad.  reduction 'mean' is the
    reference; 'sum' also returns the token count (used by the data-parallel step)."""
    tgt = shift_targets(targets, pad_idx)
    live = tgt != pad_idx
    picked = -logp.gather(2, tgt.unsqueeze(2)).squeeze(2)
    total = (picked * live.to(logp.dtype)).sum()
    n = live.sum()
    if reduction == "mean":
        return total / n.to(logp.dtype)
    return total, n


# ----------------------------------------------------------------------------------
# a9  Model.get_auxiliary_loss            seq2seq/model.py:162-164
# ----------------------------------------------------------------------------------
def auxiliary_loss(aux_logp: torch.Tensor, target_positions: torch.Tensor) -> torch.Tensor:
    return -aux_logp.gather(1, target_positions.view(-1, 1)).mean()


# ----------------------------------------------------------------------------------
# a11 Model.get_metrics                   seq2seq/model.py:117-137
# ----------------------------------------------------------------------------------
def metrics(logp: torch.Tensor, targets: torch.Tensor, pad_idx: int = 0) -> Tuple[float, float]:
    tgt = shift_targets(targets, pad_idx)
    live = tgt != pad_idx
    hit = (logp.argmax(dim=2) == tgt) & live
    acc = 100.0 * hit.sum().item() / live.sum().item()
    exact = 100.0 * (hit.sum(dim=1) == live.sum(dim=1)).sum().item() / targets.shape[0]
    return acc, exact


# ----------------------------------------------------------------------------------
# a12 optimiser step of the loop          seq2seq/train.py:67-70,110-113
# ----------------------------------------------------------------------------------
def adam_step(params: List[torch.Tensor], grads: List[torch.Tensor], exp_avg: List[torch.Tensor],
              exp_avg_sq: List[torch.Tensor], step: int, base_lr: float, beta1: float = 0.9,
              beta2: float = 0.999, eps: float = 1e-8, lr_decay: float = 0.9, lr_decay_steps: float = 20000.0):
    """One torch.optim.Adam update (no weight decay, no amsgrad) with the LambdaLR
    factor lr_decay ** (t / lr_decay_steps) where t = number of scheduler steps
    taken so far (train.py:69-70; scheduler.step() follows optimizer.step()).
    ``step`` is 1-based."""
    lr = base_lr * (lr_decay ** ((step - 1) / lr_decay_steps))
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for w, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
        w.addcdiv_(m, denom, value=-lr / bc1)
    return lr


# ----------------------------------------------------------------------------------
# the loop body as one call                seq2seq/train.py:96-114
# ----------------------------------------------------------------------------------
def loss_and_grads(p: Params, batch: dict, conditional: bool = True, auxiliary: bool = False,
                   bidirectional: bool = True, weight_target_loss: float = 0.3, pad_idx: int = 0,
                   masks=None) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], torch.Tensor]:
    """Forward + loss + backward through autograd.  Returns (loss, grads by name, logp)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    logp, aux = forward(leaves, batch["commands"], batch["cmd_lengths"], batch["world"], batch["targets"],
                        conditional, auxiliary, bidirectional, masks)
    loss = sequence_loss(logp, batch["targets"], pad_idx)
    if auxiliary:
        loss = loss + weight_target_loss * auxiliary_loss(aux, batch["target_positions"])
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return loss.detach(), grads, logp.detach()
