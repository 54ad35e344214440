"""Flat fp32 parameter / gradient storage with ``nn.Parameter`` views.

One allocation per tower (or fusion head) holds every parameter:  [ GEMM weight blocks | everything else ].
  * the q/k/v projection weights (and biases) of one attention are adjacent, so the fused [3d, d] QKV GEMM,
    its weight gradient and its bias gradient address them as one matrix / vector;
  * the "everything else" tail (biases, LayerNorm affine, embeddings) is where gradients are accumulated with
    atomics, so one memset per step clears exactly that range;
  * Adam, the RCCL all-reduce and gradient clipping see a handful of large contiguous ranges instead of hundreds
    of small tensors (sized for 288 GB HBM and per-link-bound xGMI rings: few, large messages).
``state_dict()`` is unaffected: every reference key is still an ``nn.Parameter`` (a view into the flat buffer).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
from torch import nn

ALIGN = 64  # floats (256 B): keeps every block 16-byte aligned for vector loads and the atomics' 256-B runs


def _stored(shape, pitch) -> int:
    """floats an item occupies: contiguous, or rows of `pitch` floats (first dimension = rows, the rest padded up to the pitch)"""
    return int(torch.Size(shape).numel()) if pitch is None else int(shape[0]) * int(pitch)


class Block:
    """Contiguous run of parameters: [(name, shape) | (name, shape, pitch), ...].  With a pitch the tensor's first dimension is
    stored as rows of `pitch` floats (zero padding behind the row's own elements: the patch-embedding matrix of a 14-pixel patch
    has 588 columns, the GEMM wants a multiple of 8) and the parameter is a strided view."""

    def __init__(self, items: Sequence[Tuple], kind: str):
        assert kind in ("mat", "vec")
        self.items = [(it[0], tuple(it[1]), (it[2] if len(it) > 2 else None)) for it in items]
        self.kind = kind
        self.offset = -1
        self.numel = sum(_stored(s, pitch) for _, s, pitch in self.items)


class FlatStore:
    def __init__(self, blocks: List[Block]):
        self.blocks = [b for b in blocks if b.kind == "mat"] + [b for b in blocks if b.kind == "vec"]
        off = 0
        self.index: Dict[str, Tuple] = {}          # name -> (offset, shape, row pitch or None)
        self.vec_start = None
        for b in self.blocks:
            if b.kind == "vec" and self.vec_start is None:
                self.vec_start = off
            b.offset = off
            o = off
            for name, shape, pitch in b.items:
                self.index[name] = (o, tuple(shape), pitch)
                o += _stored(shape, pitch)
            off = (o + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        if self.vec_start is None:
            self.vec_start = off
        self.master = torch.zeros(self.total, dtype=torch.float32)
        self.grad = None

    # ---- views -------------------------------------------------------------------------------
    @staticmethod
    def _item_view(buf: torch.Tensor, o: int, shape, pitch) -> torch.Tensor:
        if pitch is None:
            return buf[o:o + int(torch.Size(shape).numel())].view(shape)
        inner = [1]
        for n in reversed(shape[2:]):
            inner.insert(0, inner[0] * n)
        return buf.as_strided(shape, (pitch, *inner), buf.storage_offset() + o)

    def view(self, name: str) -> torch.Tensor:
        return self._item_view(self.master, *self.index[name])

    def gview(self, name: str) -> torch.Tensor:
        return self._item_view(self.grad, *self.index[name])

    def block_view(self, block: Block, buf: torch.Tensor) -> torch.Tensor:
        return buf[block.offset:block.offset + block.numel]

    def ensure_grad(self) -> torch.Tensor:
        if self.grad is None or self.grad.device != self.master.device:
            self.grad = torch.zeros_like(self.master)
        return self.grad

    def zero_accumulated(self):
        """clear the part of the gradient buffer that is accumulated into (biases / LayerNorm / embedding gradients: atomics).
        The GEMM weight blocks in front of it are overwritten by every backward (split-K slices meet in a workspace, the
        reduce kernel stores), so the ~340 MB of them per tower are not cleared."""
        self.ensure_grad()[self.vec_start:].zero_()

    def move(self, fn):
        new = fn(self.master)
        if new.dtype != torch.float32:
            raise TypeError("master parameters stay fp32; pick the compute dtype with set_compute_dtype()")
        moved = new.device != self.master.device or new.data_ptr() != self.master.data_ptr()
        self.master = new
        if moved:
            self.grad = None
        return moved


class Node(nn.Module):
    """Name-only container: gives parameters the reference's dotted state-dict keys; no arithmetic lives here."""

    def __getitem__(self, i):
        return getattr(self, str(i))

    def __len__(self):
        return len(self._modules)


def attach(root: nn.Module, dotted: str, param: nn.Parameter):
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, Node())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], param)


def get_param(root: nn.Module, dotted: str) -> nn.Parameter:
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        mod = mod._modules[p]
    return mod._parameters[parts[-1]]
