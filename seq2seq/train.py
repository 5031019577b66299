"""Drop-in name for seq2seq/train.py: `train(...)` with the reference's keyword arguments (train.py:15-23)."""
from multimodal_seq2seq_gscan_amd.train import train_on_dataset as train  # noqa: F401
