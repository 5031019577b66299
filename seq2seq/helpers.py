"""Drop-in name for seq2seq/helpers.py."""
from multimodal_seq2seq_gscan_amd.helpers import log_parameters, sequence_accuracy, sequence_mask  # noqa: F401
