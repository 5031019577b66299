"""Shared helpers for the test-suite (fixture loading, oracle plumbing)."""
from __future__ import annotations

import os

import numpy as np
import torch

from weights import golden_weights  # tests/golden/weights.py

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

BATCH_KEYS = ("commands", "cmd_lengths", "world", "targets", "tgt_lengths", "target_positions")


def load_fixture(name: str) -> dict:
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def fixture_batch(fx: dict) -> dict:
    return {k: torch.from_numpy(fx[k]) for k in BATCH_KEYS}


def fixture_params(cfg: dict, fx: dict, dtype=torch.float32) -> dict:
    w = golden_weights(cfg, int(fx["seed_weights"]))
    return {k: torch.from_numpy(v).to(dtype) for k, v in w.items()}


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))
