"""Data-parallel training engine: flat gradients, RCCL all-reduce overlapped with backward, fused Adam.

Replaces, for the hot path, what the reference gets from ``DistributedDataParallel`` + ``torch.optim.Adam``
(train_ddp.py:188-189,205,253-254):
  * every tower already owns ONE flat fp32 gradient buffer; the remaining parameters (projections, fusion head) are
    flattened here into one more, so a step moves a handful of large messages (xGMI rings are per-link bound: few and
    large beats DDP's 25 MB buckets);
  * as soon as a tower's hand-written backward has enqueued its last kernel, its gradient buffer is all-reduced
    (SUM) on a dedicated HIP stream while the next tower's backward runs; the 1/world_size mean is folded into Adam;
  * Adam is one launch per flat buffer (28 B/param of HBM traffic) and marks the towers' compute-dtype weight copies
    stale so they are re-derived before the next forward.
One process per GPU (``torchrun`` env), ``torch.distributed`` backend "nccl" == RCCL; "gloo" works for CPU rehearsal
of the collective logic (tests), where Adam falls back to nothing - there is no CPU product path.
"""
from __future__ import annotations

import os

from typing import List, Optional

import torch
import torch.distributed as dist
from torch import nn

from . import ops
from .towers import ClipTower


import contextlib

_null = contextlib.nullcontext
_FUSED_UPDATE = os.environ.get("MISSM_FUSED_ADAM", "1") != "0"   # A/B switch: Adam fused with the weight-shadow refresh
_BUCKET_UPDATE = os.environ.get("MISSM_BUCKET_ADAM", "1") != "0"  # A/B switch: eager mode updates bucket by bucket (else tower by tower)


class FlatGroup:
    """Flatten an arbitrary list of parameters into one fp32 buffer (+ gradient buffer) and re-point them at views."""

    def __init__(self, params: List[nn.Parameter]):
        self.params = params
        sizes = [(p.numel() + 63) // 64 * 64 for p in params]
        total = sum(sizes)
        dev = params[0].device
        self.master = torch.zeros(total, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        self.views = []
        for p, sz in zip(params, sizes):
            v = self.master[off:off + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            g = self.grad[off:off + p.numel()].view(p.shape)
            p.grad = g
            self.views.append(g)
            off += sz

    def reattach(self):
        for p, g in zip(self.params, self.views):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                if p.grad is not None:
                    g.copy_(p.grad)
                p.grad = g


def _covers(done, tower) -> bool:
    """the ranges updated during the backward are exactly the tower's layer buckets (everything but its tail range)"""
    return sorted(done) == sorted(tower.bucket_ranges()[:-1])


class TrainEngine:
    def __init__(self, model: nn.Module, lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 process_group=None, overlap: bool = True, eager_step: bool = False):
        self.model = model
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.towers: List[ClipTower] = []
        frozen_towers = []
        for m in model.modules():
            if not isinstance(m, ClipTower):
                continue
            flags = {p.requires_grad for p in m.parameters()}
            if flags == {False}:
                frozen_towers.append(m)      # a frozen tower (e.g. a distillation teacher's encoder) is simply not optimised
            elif flags == {True}:
                self.towers.append(m)
            elif m.lora and all(p.requires_grad == m.is_trainable(n) for n, p in zip(m._param_names, m._plist)):
                # LoRA tower (image/modeling_image.py:775-793): frozen encoder, trainable adapters + embeddings + outer LayerNorms -
                # the one sanctioned mix; optimizer and all-reduce touch only tower.trainable_ranges()
                self.towers.append(m)
            else:
                # the optimizer and the all-reduce address a tower as ONE flat buffer: per-parameter freezing inside a tower
                # would still be updated (weight decay / stale Adam moments) - refuse instead of doing that silently
                raise NotImplementedError("TrainEngine: a tower must be trainable or frozen as a whole (requires_grad differs "
                                          "between its parameters)")
        owned = {id(p) for t in self.towers + frozen_towers for p in t.parameters()}
        rest = [p for p in model.parameters() if id(p) not in owned and p.requires_grad]
        self.rest = FlatGroup(rest) if rest else None
        self.step_count = 0
        self._state = {}
        self._pending = []
        self.overlap = overlap and self.world > 1
        # eager_step (opt-in; bench.py and train_ddp.train turn it on): a tower's Adam update (and the refresh of its bf16
        # weight copies) is enqueued as soon as its gradient is final - right behind its backward (1 GPU) or behind its
        # all-reduce on the communication stream (N GPUs) - instead of serialising every optimizer launch after the whole
        # backward.  That means ``loss.backward()`` itself updates the towers' weights: it is only valid for loops that run
        # exactly ONE backward per tower per step (the reference's loop, train_ddp.py:249-254) and never read or clip
        # ``.grad`` before ``step()``; a second backward of a tower before ``step()`` raises.
        self.eager_step = eager_step
        self._eager_done = set()
        self._host_sync_before_collective = dist.is_initialized() and dist.get_backend(process_group) == "gloo"
        self.comm_stream = torch.cuda.Stream() if (self.world > 1 and torch.cuda.is_available()) else None
        # eager mode updates a tower bucket by bucket (three layers' weight matrices at a time, as their gradients become final)
        # on a side stream: on one GPU a stream of its own, on N GPUs the communication stream, behind the bucket's all-reduce -
        # only the last bucket's update is left behind the backward instead of the whole tower's (0.7 ms for the video tower)
        self.update_stream = (self.comm_stream if self.world > 1 else torch.cuda.Stream()) if (eager_step and torch.cuda.is_available()) else None
        self._bucket_updated = {}            # id(tower) -> [flat ranges already updated during this backward]
        for t in self.towers:
            # (LoRA towers: two small trainable ranges - reduced and updated in step(), nothing rides inside the backward)
            t._post_backward = self._tower_done if ((self.world > 1 or eager_step) and not t.lora) else None
            t._bucket_hook = self._bucket_ready if ((self.overlap or eager_step) and not t.lora) else None
        if self.world > 1:
            self.broadcast_parameters()

    # ---- flat buffers -----------------------------------------------------------------------------
    def flat_buffers(self):
        out = [(t.flat_master(), t.flat_grad(), t) for t in self.towers]
        if self.rest is not None:
            out.append((self.rest.master, self.rest.grad, None))
        return out

    def num_parameters(self) -> int:
        return sum(p.numel() for p in self.model.parameters())

    def broadcast_parameters(self, src: int = 0):
        """DDP constructor semantics (train_ddp.py:189): rank 0's parameters win."""
        for master, _, t in self.flat_buffers():
            dist.broadcast(master, src, group=self.pg)
            if t is not None:
                t.mark_dirty()

    # ---- gradient exchange ------------------------------------------------------------------------
    def _all_reduce_async(self, grad: torch.Tensor):
        if self._host_sync_before_collective and grad.is_cuda:
            torch.cuda.current_stream().synchronize()   # gloo rehearsal only: its CUDA path is not stream-ordered like RCCL's
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                h = dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        else:
            h = dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        self._pending.append(h)

    def _bucket_ready(self, tower: ClipTower, lo: int, hi: int):
        """called from inside a tower's backward: flat range [lo, hi) of its gradient is final (reverse execution order, like
        DDP's buckets - train_ddp.py:188-189,253 - but as a few 85-115 MB messages: xGMI rings are per-link bound)"""
        if self.world > 1 and self.overlap:
            if tower._accumulate:
                raise RuntimeError("TrainEngine(overlap=True): accumulating backward (see _tower_done)")
            self._all_reduce_async(tower.flat_grad()[lo:hi])
        elif self.world > 1:
            return                                # overlap=False: everything is reduced in reduce_gradients()
        if not (_FUSED_UPDATE and _BUCKET_UPDATE and self.eager_step and tower.flat_master().is_cuda and tower.fused_update_ready()) or tower._accumulate:
            return                                # (an accumulating second backward is refused in _tower_done)
        if id(tower) in self._eager_done:
            return                                # second backward before step(): _tower_done raises
        # the bucket's update: behind its reduction on the communication stream (N GPUs) / behind the kernels that produced it
        # on the update stream (1 GPU).  The layers below never read these weights again in this backward.
        hs = []
        if self.world > 1:
            hs, self._pending = self._pending, []
        else:
            self.update_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.update_stream):
            for h in hs:
                h.wait()
            self._adam_on(tower.flat_master(), tower.flat_grad(), tower, flat_range=(lo, hi))
        self._bucket_updated.setdefault(id(tower), []).append((lo, hi))

    def _tower_done(self, tower: ClipTower):
        """called by the tower's backward on the stream it ran on, after its last kernel was enqueued"""
        if id(tower) in self._eager_done:
            raise RuntimeError("TrainEngine(eager_step=True): a tower ran a second backward before step(); its weights were "
                               "already updated by the first one.  Use eager_step=False for gradient accumulation or models "
                               "that call one tower twice per step.")
        if self.world > 1 and self.overlap and tower._accumulate:
            raise RuntimeError("TrainEngine(overlap=True) all-reduces a tower's gradient right behind its backward: a second, "
                               "accumulating backward before step() would be reduced twice.  Use overlap=False for gradient "
                               "accumulation.")
        done = self._bucket_updated.pop(id(tower), [])
        rest = None if not done else (tower.tail_range() if _covers(done, tower) else "mixed")
        if rest == "mixed":
            raise RuntimeError("TrainEngine: a tower's buckets were updated only in part during its backward")
        if self.world > 1:
            if not self.overlap:
                return
            lo, hi = tower.tail_range()           # the layer buckets left during the backward (_bucket_ready); this is the rest
            self._all_reduce_async(tower.flat_grad()[lo:hi])
            if self.eager_step and tower.flat_master().is_cuda:
                # every reduction of this tower was issued from this thread, in order; waiting for all pending handles on the
                # communication stream orders the update behind them (other towers' handles are at most a few buckets)
                hs, self._pending = self._pending, []
                with torch.cuda.stream(self.comm_stream) if self.comm_stream is not None else _null():
                    for h in hs:
                        h.wait()                  # comm stream waits for the reductions, then updates (the rest of) this tower
                    self._adam_on(tower.flat_master(), tower.flat_grad(), tower, flat_range=rest)
                    tower._ensure_ready()
                self._eager_done.add(id(tower))
        elif self.eager_step and tower.flat_master().is_cuda:
            if rest is None:
                self._adam_on(tower.flat_master(), tower.flat_grad(), tower)
            else:
                self.update_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.update_stream):
                    self._adam_on(tower.flat_master(), tower.flat_grad(), tower, flat_range=rest)
            tower._ensure_ready()                 # re-derive the compute-dtype weight copies behind the update
            self._eager_done.add(id(tower))

    def zero_grad(self):
        """forget every gradient: the flattened remainder is cleared, the towers' next backward overwrites (instead of
        accumulating into) their buffers"""
        for t in self.towers:
            t._grad_fresh = False
        if self.rest is not None:
            self.rest.reattach()
            self.rest.grad.zero_()

    def reduce_gradients(self):
        """finish the gradient exchange: all-reduce (SUM) what is still local and wait for the overlapped reductions"""
        if self.world > 1:
            for t in self.towers:
                if t.lora and t._grad_fresh:
                    for lo, hi in t.trainable_ranges():
                        self._all_reduce_async(t.flat_grad()[lo:hi])
            if not self.overlap:
                for _, g, t in self.flat_buffers():
                    if t is not None and t._grad_fresh and not t.lora:
                        self._all_reduce_async(g)
            if self.rest is not None:
                self.rest.reattach()
                self._all_reduce_async(self.rest.grad)
            for h in self._pending:
                h.wait()
            self._pending.clear()
            if self.comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        elif self.rest is not None:
            self.rest.reattach()
        if self.world == 1 and self.update_stream is not None:
            torch.cuda.current_stream().wait_stream(self.update_stream)   # the towers' bucket-wise updates

    def _buffer_index(self, t) -> int:
        """position of a tower's (or, for None, the flattened remainder's) buffer in flat_buffers() - the key of its Adam moments"""
        if t is None:
            return len(self.towers)
        for i, x in enumerate(self.towers):
            if x is t:
                return i
        raise KeyError("tower is not optimised by this engine")

    def _moments(self, master, t):
        """Adam moments of one flat buffer, keyed by the buffer's POSITION (keyed by ``master.data_ptr()`` a rebind of the
        tower - ``model.to(device)``, ``set_compute_dtype`` - silently dropped restored moments while the step count kept its value).
        Moments follow the buffer to its device; a buffer whose size changed is an error, not a restart from zero."""
        key = self._buffer_index(t)
        mv = self._state.get(key)
        if mv is None:
            mv = self._state[key] = (torch.zeros_like(master), torch.zeros_like(master))
        elif mv[0].numel() != master.numel():
            raise RuntimeError("TrainEngine: the optimizer state of a flat buffer does not match its size (model changed after load_state_dict?)")
        elif mv[0].device != master.device:
            mv = self._state[key] = (mv[0].to(master.device), mv[1].to(master.device))
        return mv

    def _adam_on(self, master, grad, t, flat_range=None):
        if not master.is_cuda:
            raise RuntimeError("TrainEngine.apply_adam: parameters are not on a GPU (no CPU optimizer path)")
        m, v = self._moments(master, t)
        hp = (self.step_count + 1, self.lr, self.betas[0], self.betas[1], self.eps, self.wd)
        if t is not None and t.lora:
            # adapter-only training: Adam on the trainable ranges; the frozen encoder (and its gradient scratch) is never touched;
            # the merged GEMM weights W + (alpha / r) B A are rebuilt before the next forward (ClipTower._refresh_shadows)
            for lo, hi in t.trainable_ranges():
                ops.adam_step(master[lo:hi], grad[lo:hi], m[lo:hi], v[lo:hi], *hp, grad_scale=1.0 / self.world)
            t.mark_dirty()
            t._grad_fresh = False
            return
        if flat_range is not None:
            # one gradient bucket of a tower: the weight matrices that start inside it, and - if it is the tail range (it ends
            # with the flat buffer) - the vector parameters behind the matrices
            lo, hi = flat_range
            t.adam_and_refresh(grad, m, v, *hp, grad_scale=1.0 / self.world, flat_range=flat_range)
            vs = max(lo, t.vec_start())
            if hi == master.numel() and vs < hi:
                ops.adam_step(master[vs:], grad[vs:], m[vs:], v[vs:], *hp, grad_scale=1.0 / self.world)
                t._grad_fresh = False
            return
        if _FUSED_UPDATE and t is not None and t.fused_update_ready():
            # weight matrices: Adam fused with the refresh of their compute-dtype copies (one pass over p, g, m, v);
            # everything else (biases, LayerNorm, embeddings: the tail of the flat buffer): the plain fused Adam
            t.adam_and_refresh(grad, m, v, *hp, grad_scale=1.0 / self.world)
            lo = t.vec_start()
            if lo < master.numel():
                ops.adam_step(master[lo:], grad[lo:], m[lo:], v[lo:], *hp, grad_scale=1.0 / self.world)
            t._grad_fresh = False
            return
        ops.adam_step(master, grad, m, v, *hp, grad_scale=1.0 / self.world)
        if t is not None:
            t.mark_dirty()
            t._grad_fresh = False

    def apply_adam(self):
        """one fused Adam launch per flat buffer (towers already updated eagerly are skipped); the 1/world mean of the SUM
        all-reduce is folded in"""
        for master, grad, t in self.flat_buffers():
            if t is not None and (id(t) in self._eager_done or not t._grad_fresh):
                t._grad_fresh = False
                continue   # updated eagerly, or not used this step (e.g. no 'language' input: gradient None in the reference)
            self._adam_on(master, grad, t)
        self._eager_done.clear()
        self.step_count += 1

    def step(self):
        self.reduce_gradients()
        self.apply_adam()

    # ---- optimizer state (checkpoint 'optimizer_state_dict', reference train_ddp.py:298-306) -----------------------
    def _located_params(self):
        """[(name, parameter, flat-buffer index or None, element offset)] in ``model.named_parameters()`` order - the order in which
        ``optim.Adam(model.parameters())`` (reference train_ddp.py:205) numbers its state"""
        bufs = [(m.data_ptr(), m.data_ptr() + 4 * m.numel()) for m, _, _ in self.flat_buffers()]
        out = []
        for name, p in self.model.named_parameters():
            where = None
            for i, (lo, hi) in enumerate(bufs):
                if lo <= p.data_ptr() < hi:
                    where = (i, (p.data_ptr() - lo) // 4)
                    break
            out.append((name, p, where))
        return out

    def state_dict(self):
        """``torch.optim.Adam.state_dict()`` layout (what the reference saves, train_ddp.py:303): ``{'state': {i: {'step', 'exp_avg',
        'exp_avg_sq'}}, 'param_groups': [{lr, betas, eps, weight_decay, amsgrad, ..., 'params': [0..n-1], 'param_names': [...]}]}``
        with ``i`` numbering ``model.parameters()``; every per-parameter moment is the parameter-shaped slice of the flat moment
        buffers (padding rows of the padded patch matrix are not part of it).  ``param_names`` (as recent torch writes for
        optimizers built from named parameters) lets a loader match by name when module registration order differs."""
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        bufs = self.flat_buffers()
        state, names = {}, []
        for i, (name, p, where) in enumerate(self._located_params()):
            names.append(name)
            if where is None or not p.requires_grad:     # (frozen parameters have no optimizer state, as in torch)
                continue
            mv = self._state.get(where[0])
            if mv is None or self.step_count == 0:
                continue
            master = bufs[where[0]][0]
            off = where[1] + mv[0].storage_offset()
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": mv[0].as_strided(p.shape, p.stride(), off).detach().cpu().contiguous().clone(),
                        "exp_avg_sq": mv[1].as_strided(p.shape, p.stride(), off).detach().cpu().contiguous().clone()}
            assert master.numel() == mv[0].numel()
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "params": list(range(len(names))), "param_names": names}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """accepts ``torch.optim.Adam.state_dict()`` (with ``param_names``: matched by name; without: by position, shapes checked) and
        the round-2 private layout (``{'flat': [...]}``)"""
        bufs = self.flat_buffers()
        if "flat" in sd:                              # round-2 checkpoints
            if len(sd["flat"]) != len(bufs):
                raise ValueError("optimizer state does not match this model's flat buffers")
            self.step_count, self.lr, self.betas, self.eps, self.wd = sd["step"], sd["lr"], tuple(sd["betas"]), sd["eps"], sd["weight_decay"]
            for i, ((master, _, _), st) in enumerate(zip(bufs, sd["flat"])):
                if st is None:
                    self._state.pop(i, None)
                    continue
                if st["exp_avg"].numel() != master.numel():
                    raise ValueError("optimizer state does not match this model's flat buffers")
                self._state[i] = (st["exp_avg"].to(master.device, torch.float32).clone(), st["exp_avg_sq"].to(master.device, torch.float32).clone())
            return
        if "state" not in sd or "param_groups" not in sd or len(sd["param_groups"]) != 1:
            raise ValueError("optimizer state: expected torch.optim.Adam's {'state', 'param_groups'} with one parameter group")
        grp = sd["param_groups"][0]
        if grp.get("amsgrad"):
            raise ValueError("optimizer state: amsgrad is not implemented by the fused Adam kernel")
        located = self._located_params()
        ids = list(grp["params"])
        names = grp.get("param_names")
        if names is not None:
            by_name = dict(zip(names, ids))
            missing = [n for n, _, w in located if w is not None and n not in by_name]
            if missing:
                raise ValueError(f"optimizer state lacks parameters {missing[:4]}{' ...' if len(missing) > 4 else ''}")
            pick = [by_name.get(n) for n, _, _ in located]
        else:
            if len(ids) != len(located):
                raise ValueError(f"optimizer state numbers {len(ids)} parameters, the model has {len(located)}")
            pick = ids
        self.lr, self.betas, self.eps, self.wd = grp["lr"], tuple(grp["betas"]), grp["eps"], grp["weight_decay"]
        self._state = {}
        steps = set()
        for (name, p, where), idx in zip(located, pick):
            st = sd["state"].get(idx) if idx is not None else None
            if st is None or where is None or not p.requires_grad:
                continue
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state of {name}: shape {tuple(st['exp_avg'].shape)} vs parameter {tuple(p.shape)}")
            master = bufs[where[0]][0]
            if where[0] not in self._state:
                self._state[where[0]] = (torch.zeros_like(master), torch.zeros_like(master))
            m, v = self._state[where[0]]
            off = where[1] + m.storage_offset()
            m.as_strided(p.shape, p.stride(), off).copy_(st["exp_avg"].to(m.device, torch.float32))
            v.as_strided(p.shape, p.stride(), off).copy_(st["exp_avg_sq"].to(v.device, torch.float32))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"optimizer state: parameters disagree on the step count {sorted(steps)} (one fused step per flat buffer)")
        self.step_count = steps.pop() if steps else 0
