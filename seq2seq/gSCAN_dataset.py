"""Drop-in name for seq2seq/gSCAN_dataset.py (Vocabulary :17-102, GroundedScanDataset :105-310)."""
from multimodal_seq2seq_gscan_amd.dataset import GroundedScanDataset, Vocabulary  # noqa: F401
