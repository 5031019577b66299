"""The small helpers of seq2seq/helpers.py, for code that imports them from the drop-in package."""
from __future__ import annotations

import logging

import torch

from .predict import sequence_accuracy  # noqa: F401  (helpers.py:44-64)

logger = logging.getLogger(__name__)


def sequence_mask(sequence_lengths: torch.Tensor, max_len=None) -> torch.Tensor:
    """helpers.py:11-32: [batch, max_len] boolean mask, True where the position index is below the row's length."""
    if max_len is None:
        max_len = int(sequence_lengths.max())
    positions = torch.arange(int(max_len), device=sequence_lengths.device, dtype=torch.long)
    return positions.unsqueeze(0) < sequence_lengths.to(torch.long).unsqueeze(1)


def log_parameters(model: torch.nn.Module) -> None:
    """helpers.py:35-41: total and per-tensor parameter sizes to the log."""
    trainable = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    logger.info("Total parameters: %d" % sum(p.numel() for _, p in trainable))
    for name, p in trainable:
        logger.info("%s : %s" % (name, list(p.size())))
