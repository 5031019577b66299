"""ctypes binding of libgscan_hip.so (C ABI: include/gscan_hip.h).

There is no CPU fallback: if the shared object is missing or does not export the ABI the
package was written against, loading raises and every product entry point fails loudly."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
# GSCAN_HIP_LIB: development override, used by tools/variants.py to time experimental builds side by side
LIB_PATH = os.environ.get("GSCAN_HIP_LIB") or os.path.join(HERE, "libgscan_hip.so")
ABI_VERSION = 14
MAX_ENC_LAYERS = 4
COMM_ID_BYTES = 128

_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)

DIM_FIELDS = ("B", "L", "T", "G", "C", "Co", "K3", "E", "He", "H", "Vi", "V", "conditional", "auxiliary",
              "bidirectional", "pad_in", "pad_tgt", "enc_layers")

# (C field, reference state_dict name) in named_parameters() order; see include/gscan_hip.h
PARAM_FIELDS = (
    ("conv1_w", "situation_encoder.conv_1.weight"), ("conv1_b", "situation_encoder.conv_1.bias"),
    ("conv2_w", "situation_encoder.conv_2.weight"), ("conv2_b", "situation_encoder.conv_2.bias"),
    ("conv3_w", "situation_encoder.conv_3.weight"), ("conv3_b", "situation_encoder.conv_3.bias"),
    ("vis_key_w", "visual_attention.key_layer.weight"), ("vis_query_w", "visual_attention.query_layer.weight"),
    ("vis_energy_w", "visual_attention.energy_layer.weight"),
    ("enc_emb", "encoder.embedding.weight"),
    ("enc_w_ih", "encoder.lstm.weight_ih_l0"), ("enc_w_hh", "encoder.lstm.weight_hh_l0"),
    ("enc_b_ih", "encoder.lstm.bias_ih_l0"), ("enc_b_hh", "encoder.lstm.bias_hh_l0"),
    ("enc_w_ih_rev", "encoder.lstm.weight_ih_l0_reverse"), ("enc_w_hh_rev", "encoder.lstm.weight_hh_l0_reverse"),
    ("enc_b_ih_rev", "encoder.lstm.bias_ih_l0_reverse"), ("enc_b_hh_rev", "encoder.lstm.bias_hh_l0_reverse"),
    ("bridge_w", "enc_hidden_to_dec_hidden.weight"), ("bridge_b", "enc_hidden_to_dec_hidden.bias"),
    ("txt_key_w", "textual_attention.key_layer.weight"), ("txt_query_w", "textual_attention.query_layer.weight"),
    ("txt_energy_w", "textual_attention.energy_layer.weight"),
    ("q2k_w", "attention_decoder.queries_to_keys.weight"), ("q2k_b", "attention_decoder.queries_to_keys.bias"),
    ("dec_emb", "attention_decoder.embedding.weight"),
    ("dec_w_ih", "attention_decoder.lstm.weight_ih_l0"), ("dec_w_hh", "attention_decoder.lstm.weight_hh_l0"),
    ("dec_b_ih", "attention_decoder.lstm.bias_ih_l0"), ("dec_b_hh", "attention_decoder.lstm.bias_hh_l0"),
    ("out2hid_w", "attention_decoder.output_to_hidden.weight"),
    ("hid2out_w", "attention_decoder.hidden_to_output.weight"),
)


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in DIM_FIELDS]


# encoder layers 1..: (slot in gscan_params.enc_deep[layer-1], state_dict name pattern)
ENC_DEEP_FIELDS = ("weight_ih_l{}", "weight_hh_l{}", "bias_ih_l{}", "bias_hh_l{}", "weight_ih_l{}_reverse",
                   "weight_hh_l{}_reverse", "bias_ih_l{}_reverse", "bias_hh_l{}_reverse")


class Params(C.Structure):
    _fields_ = [(n, C.c_void_p) for n, _ in PARAM_FIELDS] + [("enc_deep", (C.c_void_p * 8) * (MAX_ENC_LAYERS - 1))]


class Batch(C.Structure):
    _fields_ = [("commands", C.c_void_p), ("cmd_lengths", C.c_void_p), ("world", C.c_void_p),
                ("targets", C.c_void_p), ("target_positions", C.c_void_p), ("world_u8", C.c_void_p)]


class Masks(C.Structure):
    _fields_ = [("cnn", C.c_void_p), ("enc", C.c_void_p), ("dec", C.c_void_p), ("enc_deep", C.c_void_p),
                ("in_kernel", C.c_int32), ("p_cnn", C.c_float), ("p_enc", C.c_float), ("p_dec", C.c_float),
                ("seed", C.c_uint64), ("stream_id", C.c_uint64)]


_vp, _i, _f, _sz, _i64, _u64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_int64, C.c_uint64

# name -> (restype, argtypes); every symbol include/gscan_hip.h declares
PROTOTYPES = {
    "gscan_abi_version": (_i, []),
    "gscan_last_error": (C.c_char_p, []),
    "gscan_workspace_bytes": (_sz, [C.POINTER(Dims)]),
    "gscan_decoder_kernel_family": (_i, [C.POINTER(Dims)]),
    "gscan_workspace_find": (_i, [C.POINTER(Dims), C.c_char_p, C.POINTER(_sz), C.POINTER(_sz)]),
    "gscan_forward": (_i, [C.POINTER(Dims), C.POINTER(Params), C.POINTER(Batch), C.POINTER(Masks), _vp, _vp, _vp, _vp]),
    "gscan_backward": (_i, [C.POINTER(Dims), C.POINTER(Params), C.POINTER(Batch), C.POINTER(Masks), _vp, _vp, _vp,
                            C.POINTER(Params), _vp]),
    "gscan_sequence_nll": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "gscan_position_nll": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "gscan_sequence_metrics": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "gscan_adam_step": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _f, _i64, _vp, _vp]),
    "gscan_adam_step_zero_grad": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _f, _i64, _vp, _vp]),
    "gscan_dropout_mask": (_i, [_vp, _sz, _f, _u64, _u64, _vp]),
    "gscan_step_losses": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "gscan_loss_seeds": (_i, [_vp, _f, _i, _vp, _vp]),
    "gscan_encode": (_i, [C.POINTER(Dims), C.POINTER(Params), C.POINTER(Batch), C.POINTER(Masks), _vp, _vp]),
    "gscan_decode_step": (_i, [C.POINTER(Dims), C.POINTER(Params), C.POINTER(Batch), _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp, _vp, _vp]),
    "gscan_decode_batched": (_i, [C.POINTER(Dims), C.POINTER(Params), C.POINTER(Batch), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "gscan_greedy_decode": (_i, [C.POINTER(Dims), _i, C.POINTER(Params), C.POINTER(Batch), _vp, _i, _i, _vp, _vp, _vp, _vp,
                                 _vp, _vp]),
    "gscan_backward_nll": (_i, [C.POINTER(Dims), C.POINTER(Params), C.POINTER(Batch), C.POINTER(Masks), _vp, _f, _i,
                                _vp, _vp, C.POINTER(Params), _vp]),
    "gscan_train_step_nll": (_i, [C.POINTER(Dims), C.POINTER(Params), C.POINTER(Batch), C.POINTER(Masks), _vp, _vp, _vp, _f,
                                  _i, _vp, _vp, C.POINTER(Params), _vp]),
    "gscan_adam_step_mean": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _f, _i64, _vp, _vp]),
    "gscan_backward_seeded": (_i, [C.POINTER(Dims), C.POINTER(Params), C.POINTER(Batch), C.POINTER(Masks), _vp, _vp,
                                   _vp, _vp, C.POINTER(Params), _vp]),
    "gscan_adam_step_masks": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _f, _i64, _vp, _vp, _sz, _sz, _sz, _f, _f, _f,
                                   _u64, _u64, _vp]),
    "gscan_trace_set": (_i, [_vp]),
    "gscan_dropout_masks": (_i, [_vp, _sz, _sz, _sz, _f, _f, _f, _u64, _u64, _vp, _vp]),
    "gscan_dropout_masks_kernel_layout": (_i, [_vp, _vp, _vp, _vp, _f, _f, _f, _u64, _u64, _vp]),
    "gscan_comm_available": (_i, []),
    "gscan_comm_unique_id": (_i, [_vp]),
    "gscan_comm_init": (_i, [C.POINTER(_vp), _i, _i, _vp]),
    "gscan_allreduce_f32": (_i, [_vp, _vp, _sz, _vp]),
    "gscan_comm_destroy": (_i, [_vp]),
    "gscan_comm_count": (_i, [_vp, C.POINTER(_i)]),
    "gscan_early_gradients_wait": (_i, [_vp]),
    "gscan_comm_set_early_allreduce": (_i, [_vp, _vp, _sz]),
    "gscan_probe_enable": (_i, [_i]),
    "gscan_probe_reset": (_i, []),
    "gscan_probe_read": (_i, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                              C.POINTER(_i64)]),
    "gscan_gemm_f32": (_i, [_i, _i, _i, _f, _vp, _i64, _i64, _vp, _i64, _i64, _f, _vp, _i64, _vp, _i, _vp, _i, _vp]),
    "gscan_gemm_f32_scratch": (_i, [_i, _i, _i, _f, _vp, _i64, _i64, _vp, _i64, _i64, _f, _vp, _i64, _vp, _i, _vp, _i, _vp,
                                    _vp, _sz, _vp]),
    "gscan_world_encoder_forward": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_vp), _i, _i, _i, _i, _i, _vp, _vp, _vp,
                                         _vp]),
    "gscan_world_encoder_backward_scratch_floats": (_sz, [_i, _i, _i]),
    "gscan_world_encoder_backward": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "gscan_encoder_lstm_forward": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                        _vp]),
    "gscan_encoder_lstm_backward": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib: Optional[C.CDLL] = None


class GscanError(RuntimeError):
    """A call into libgscan_hip.so returned non-zero."""


def load() -> C.CDLL:
    """Load the library once.  torch must be imported first so that both share one HIP runtime
    (the wheel bundles libamdhip64.so.7; the loader resolves our NEEDED entry to that copy)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (loads the HIP runtime the process will use)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP kernels are not built and there is no CPU fallback. "
            "Run `python -m multimodal_seq2seq_gscan_amd.build` (needs hipcc).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.gscan_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} has ABI version {got}, this package needs {ABI_VERSION}; rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise GscanError(f"{what}: {load().gscan_last_error().decode(errors='replace')}")


def ptr(t) -> Optional[int]:
    """Device (or host) address of a tensor, or None."""
    return None if t is None else t.data_ptr()
