"""Deterministic, platform-stable parameter values for the golden fixtures.

Fixtures do not store the (large) weight tensors; both the generating script (which feeds
them to the imported reference) and the tests (which feed them to the oracle and to the
HIP path) rebuild them from this recipe with numpy's PCG64 generator."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np


def parameter_shapes(cfg: dict) -> "OrderedDict[str, Tuple[int, ...]]":
    """Parameter names and shapes of the reference model, in ``named_parameters()`` order
    (README.md:265-296 lists them for the demo; seq2seq/model.py:47-87 is the construction order)."""
    C, Co, k = cfg["num_cnn_channels"], cfg["cnn_hidden_num_channels"], cfg["cnn_kernel_size"]
    E, He, H = cfg["embedding_dimension"], cfg["encoder_hidden_size"], cfg["decoder_hidden_size"]
    Vi, V = cfg["input_vocabulary_size"], cfg["target_vocabulary_size"]
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for name, kk in (("conv_1", 1), ("conv_2", 5), ("conv_3", k)):
        s[f"situation_encoder.{name}.weight"] = (Co, C, kk, kk)
        s[f"situation_encoder.{name}.bias"] = (Co,)
    s["visual_attention.key_layer.weight"] = (H, 3 * Co)
    s["visual_attention.query_layer.weight"] = (H, H)
    s["visual_attention.energy_layer.weight"] = (1, H)
    s["encoder.embedding.weight"] = (Vi, E)
    D = 2 if cfg["encoder_bidirectional"] else 1
    for layer in range(int(cfg.get("num_encoder_layers", 1))):         # nn.LSTM order: layer, then direction
        for suffix in ([""] + (["_reverse"] if cfg["encoder_bidirectional"] else [])):
            s[f"encoder.lstm.weight_ih_l{layer}{suffix}"] = (4 * He, E if layer == 0 else D * He)
            s[f"encoder.lstm.weight_hh_l{layer}{suffix}"] = (4 * He, He)
            s[f"encoder.lstm.bias_ih_l{layer}{suffix}"] = (4 * He,)
            s[f"encoder.lstm.bias_hh_l{layer}{suffix}"] = (4 * He,)
    s["enc_hidden_to_dec_hidden.weight"] = (H, He)
    s["enc_hidden_to_dec_hidden.bias"] = (H,)
    s["textual_attention.key_layer.weight"] = (H, He)
    s["textual_attention.query_layer.weight"] = (H, H)
    s["textual_attention.energy_layer.weight"] = (1, H)
    if cfg["conditional_attention"]:
        s["attention_decoder.queries_to_keys.weight"] = (H, 2 * H)
        s["attention_decoder.queries_to_keys.bias"] = (H,)
    s["attention_decoder.embedding.weight"] = (V, H)
    s["attention_decoder.lstm.weight_ih_l0"] = (4 * H, 3 * H)
    s["attention_decoder.lstm.weight_hh_l0"] = (4 * H, H)
    s["attention_decoder.lstm.bias_ih_l0"] = (4 * H,)
    s["attention_decoder.lstm.bias_hh_l0"] = (4 * H,)
    s["attention_decoder.output_to_hidden.weight"] = (H, 4 * H)
    s["attention_decoder.hidden_to_output.weight"] = (V, H)
    return s


def golden_weights(cfg: dict, seed: int) -> Dict[str, np.ndarray]:
    """float32 values: matrices uniform in +-1/sqrt(fan_in), vectors +-0.1, embeddings +-1 with
    the padding row zeroed (as nn.Embedding(padding_idx=...) initialises it)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out: Dict[str, np.ndarray] = {}
    for name, shape in parameter_shapes(cfg).items():
        if "embedding" in name:
            w = rng.uniform(-1.0, 1.0, size=shape)
            pad = cfg["input_padding_idx"] if name.startswith("encoder.") else cfg["target_pad_idx"]
            w[pad] = 0.0
        elif len(shape) == 1:
            w = rng.uniform(-0.1, 0.1, size=shape)
        else:
            a = 1.0 / np.sqrt(float(np.prod(shape[1:])))
            w = rng.uniform(-a, a, size=shape)
        out[name] = w.astype(np.float32)
    return out
