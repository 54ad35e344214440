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


def synth_text_batch(batch: int, ctx: int = 77, seed: int = 0, vocab: int = 49408):
    """CLIP-tokenizer-shaped synthetic text: BOS first, ids uniform in [1000, 40000), EOS (the
    maximum id, also the pad id - tokenization_image.py:66) at position len-1 and beyond, len ~ U{8..ctx}.
    Returns (input_ids int64 [B, ctx], attention_mask int64 [B, ctx])."""
    g = torch.Generator().manual_seed(seed)
    bos, eos = vocab - 2, vocab - 1
    lo, hi = (1000, 40000) if vocab > 40000 else (1, max(2, vocab - 2))
    ids = torch.randint(lo, hi, (batch, ctx), generator=g)
    lens = torch.randint(min(8, ctx), ctx + 1, (batch,), generator=g)
    pos = torch.arange(ctx)[None]
    ids[:, 0] = bos
    ids = torch.where(pos >= (lens[:, None] - 1), torch.full_like(ids, eos), ids)
    return ids, (pos < lens[:, None]).to(torch.int64)
