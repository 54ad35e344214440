"""Synthetic inputs for the benchmark / tests (the reference's loaders read real datasets: out of scope)."""
from __future__ import annotations

import random
from typing import Sequence

import torch

from .src.model.baseline import missing_type_index


def synth_missing_index(n: int, modality_types: Sequence[str], ratio: float, seed: int) -> torch.Tensor:
    """One missing-modality code per sample, the way the reference prepares them offline
    (src/utils/generate_missing.py:21-38, 'mixed'): int(n*ratio) samples picked with random.sample, each assigned a code
    chosen uniformly among the present modalities; everything else 0 (= nothing missing)."""
    rng = random.Random(seed)
    codes = [missing_type_index[m] for m in modality_types]
    out = [0] * n
    for i in rng.sample(range(n), int(n * ratio)):
        out[i] = rng.choice(codes)
    return torch.tensor(out, dtype=torch.int64)
