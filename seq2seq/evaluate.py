"""Drop-in name for seq2seq/evaluate.py."""
from multimodal_seq2seq_gscan_amd.predict import evaluate  # noqa: F401
