"""`seq2seq.model.Model` of the reference (seq2seq/model.py:23) -> the HIP-backed drop-in."""
from multimodal_seq2seq_gscan_amd.model import Model  # noqa: F401
