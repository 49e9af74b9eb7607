"""Loader-coverage figures of the training log block (reference: utils/statistics.py:57-122, logged at
framework/train.py:371-377): how evenly a batch's rows are spread over the cache's examples and over the token positions."""

from __future__ import annotations

import math

import torch
from torch import Tensor


def _spread(indices: Tensor, support: int) -> tuple[float, float, float]:
    """(entropy in nats of the empirical distribution, entropy / log(support), distinct values / support)."""
    counts = torch.bincount(indices.reshape(-1).to(torch.int64).cpu()).to(torch.float64)
    counts = counts[counts > 0]
    if counts.numel() == 0:
        return 0.0, 0.0, 0.0
    p = counts / counts.sum()
    ent = float(-(p * p.log()).sum())
    return ent, (ent / math.log(support) if support > 1 else 0.0), counts.numel() / support


def batch_entropy(example_idx: Tensor, token_idx: Tensor, n_examples: int, content_tokens_per_example: int) -> dict[str, float]:
    """``loader/{example,token}_{entropy,entropy_normalized,coverage}`` for one batch of (example, token) indices."""
    if n_examples <= 0 or content_tokens_per_example <= 0:
        raise ValueError(f"supports must be positive, got {n_examples} examples x {content_tokens_per_example} tokens")
    if example_idx.ndim != 1 or token_idx.ndim != 1 or example_idx.shape != token_idx.shape or example_idx.numel() == 0:
        raise ValueError(f"index vectors must be 1-D, equally long and non-empty, got {tuple(example_idx.shape)} and {tuple(token_idx.shape)}")
    out: dict[str, float] = {}
    for name, idx, support in (("example", example_idx, n_examples), ("token", token_idx, content_tokens_per_example)):
        ent, norm, cov = _spread(idx, support)
        out[f"loader/{name}_entropy"] = ent
        out[f"loader/{name}_entropy_normalized"] = norm
        out[f"loader/{name}_coverage"] = cov
    return out
