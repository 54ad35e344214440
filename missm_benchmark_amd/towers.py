"""CLIP vision / text towers of LanguageBind on the HIP kernels (host-side orchestration, Python like the reference).

Mirrors ``CLIPVisionTransformer`` / ``CLIPTextTransformer`` of the reference
(languagebind/image/modeling_image.py:458-532,596-672; languagebind/video/modeling_video.py:702-784): same
constructor config fields, same ``forward`` keywords, same ``(last_hidden_state, pooled_output)`` tuple, same
``ValueError`` texts and the same state-dict keys (incl. the misspelt ``pre_layrnorm``).  The arithmetic is NOT torch:
every op is a launch into ``libmissm_hip.so``; torch only owns the buffers and the stream.

Precision: parameters, the residual stream, LayerNorm statistics and parameter gradients are fp32.  GEMM / attention
operands are ``compute_dtype`` (bf16 for throughput, fp32 for the 1e-3 parity gate) - one kernel source, two
instantiations.  The backward is hand-written (no autograd inside a tower): one ``autograd.Function`` per tower.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import List, Optional, Tuple

import torch
from torch import nn

from . import _lib, ops
from .flat import Block, FlatStore, Node, attach

_GRAD_MODE = "direct"  # "direct": kernels write parameter gradients in place; "autograd": returned through autograd (DDP-wrap safe)


def set_grad_mode(mode: str):
    global _GRAD_MODE
    assert mode in ("direct", "autograd")
    _GRAD_MODE = mode


@dataclass
class TowerConfig:
    kind: str = "vision"                 # "vision" | "text"
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    layer_norm_eps: float = 1e-5
    hidden_act: str = "quick_gelu"
    # vision (configuration_image.py:181-232)
    num_channels: int = 3
    image_size: object = 224             # int, or (height, width): the audio tower's spectrogram image (num_mel_bins, target_length)
    patch_size: int = 16
    add_time_attn: bool = False
    num_frames: int = 1
    temporal_mlp: bool = False
    force_patch_dropout: float = 0.0     # PatchDropout (image/modeling_image.py:30-63): identity at 0 / in eval; in training a random subset of the patch tokens
    # LoRA adapters over the vision encoder (configuration_image.py:200-202, image/modeling_image.py:775-793); 0 = plain weights
    lora_r: int = 0
    lora_alpha: float = 16.0
    lora_dropout: float = 0.0
    # text (configuration_image.py:70-105)
    vocab_size: int = 49408
    max_position_embeddings: int = 77

    @property
    def image_hw(self):
        hw = self.image_size
        return (int(hw), int(hw)) if isinstance(hw, int) else (int(hw[0]), int(hw[1]))

    @property
    def grid(self):
        """patch grid (rows, columns) - reference resize_pos, image/modeling_image.py:804"""
        h, w = self.image_hw
        return h // self.patch_size, w // self.patch_size

    @property
    def seq_len(self) -> int:
        return self.grid[0] * self.grid[1] + 1 if self.kind == "vision" else self.max_position_embeddings


def _r64(n: int) -> int:
    return (n + 63) // 64 * 64


class PatchDropout:
    """``PatchDropout`` of the reference (image/modeling_image.py:19-63, https://arxiv.org/abs/2212.00794): in training, with
    probability parameter ``prob`` > 0, every sample keeps its class token and ``max(1, int(num_tokens * (1 - prob)))`` patch tokens
    chosen by ``torch.randn(batch, num_tokens).topk(k)`` ON THE CPU with torch's global generator (T > 1: one draw per sample,
    shared by its frames).  ``keep_indices`` makes exactly that draw, so a run seeded like the reference drops the same tokens; a
    caller can also hand the indices in (``tower(pixel_values, patch_keep=idx)``) - which is how the parity tests pin the choice."""

    def __init__(self, prob: float, exclude_first_token: bool = True):
        assert 0 <= prob < 1.0
        self.prob = prob
        self.exclude_first_token = exclude_first_token

    def num_keep(self, num_tokens: int) -> int:
        return max(1, int(num_tokens * (1 - self.prob)))

    def keep_indices(self, B: int, T: int, num_tokens: int) -> torch.Tensor:
        rand = torch.randn(B * T if T == 1 else B, num_tokens)
        return rand.topk(self.num_keep(num_tokens), dim=-1).indices


class ClipTower(nn.Module):
    def __init__(self, config: TowerConfig, compute_dtype: torch.dtype = torch.bfloat16, seed: Optional[int] = None):
        super().__init__()
        c = config
        if c.hidden_act not in ops.ACT_CODE:
            raise ValueError(f"unsupported hidden_act {c.hidden_act!r}")
        if c.temporal_mlp and not c.add_time_attn:
            raise ValueError("temporal_mlp is the MLP of the time branch: it needs add_time_attn=True (image/modeling_image.py:73-84)")
        if c.hidden_size % c.num_attention_heads or (c.hidden_size // c.num_attention_heads) % 8:
            raise ValueError("head_dim must be a multiple of 8")
        if c.kind == "vision" and (c.image_hw[0] % c.patch_size or c.image_hw[1] % c.patch_size or c.image_hw[1] % 4):
            raise ValueError("patch_size must divide image_size (and the image width must be a multiple of 4)")
        if c.seq_len > 256 and c.hidden_size // c.num_attention_heads != 64:
            raise NotImplementedError(f"{c.seq_len} tokens per sequence need head_dim 64 (the key-chunked attention kernels)")
        if c.kind != "vision" and c.seq_len > 256:
            raise NotImplementedError("causal attention over more than 256 tokens is not instantiated")
        if not 0.0 <= c.force_patch_dropout < 1.0:
            raise ValueError("force_patch_dropout must be in [0, 1)")        # the reference's assert, image/modeling_image.py:26
        # LoRA (image/modeling_image.py:775-793): peft wraps vision_model.encoder - every encoder parameter is frozen, the target
        # linears get trainable rank-r adapters A [r, in] / B [out, r]; embeddings, pre_layrnorm and post_layernorm (outside the
        # encoder) stay trainable.  The text tower is never wrapped.
        self.lora = c.kind == "vision" and c.lora_r > 0
        if self.lora and c.lora_dropout > 0:
            raise NotImplementedError("lora_dropout > 0: the adapters ride in the merged GEMM weight W + (alpha / r) B A, which is the "
                                      "unmerged forward only without dropout on the adapter input (the reference default is 0.0)")
        self.config = c
        self.compute_dtype = compute_dtype
        self.patch_dropout = PatchDropout(c.force_patch_dropout)
        d, f = c.hidden_size, c.intermediate_size
        blocks: List[Block] = []
        patch_block = None
        if c.kind == "vision":
            # (appended AFTER the layers' matrices: the backward finishes with the embeddings, so the last gradient bucket -
            #  patch embedding + every vector parameter - is one contiguous range of the flat buffer)
            # (rows of the patch matrix padded to a multiple of 8 floats: a 14-pixel patch has 3 * 14 * 14 = 588 columns, the GEMM
            #  operands need 16-byte rows; the parameter is a strided view, the padding stays zero)
            kp = c.num_channels * c.patch_size ** 2
            patch_block = Block([("embeddings.patch_embedding.weight", (d, c.num_channels, c.patch_size, c.patch_size),
                                  None if kp % 8 == 0 else (kp + 7) // 8 * 8)], "mat")
            blocks.append(Block([("embeddings.class_embedding", (d,)), ("embeddings.position_embedding.weight", (c.seq_len, d)),
                                 ("pre_layrnorm.weight", (d,)), ("pre_layrnorm.bias", (d,))], "vec"))
        else:
            blocks.append(Block([("embeddings.token_embedding.weight", (c.vocab_size, d)),
                                 ("embeddings.position_embedding.weight", (c.max_position_embeddings, d))], "vec"))
        self._mat_blocks = {}
        for i in range(c.num_hidden_layers):
            p = f"encoder.layers.{i}"
            vec = []
            if c.add_time_attn:
                b = Block([(f"{p}.temporal_attn.{n}_proj.weight", (d, d)) for n in "qkv"], "mat"); blocks.append(b)
                self._mat_blocks[f"{p}.tqkv"] = b
                b = Block([(f"{p}.temporal_attn.out_proj.weight", (d, d))], "mat"); blocks.append(b)
                self._mat_blocks[f"{p}.tout"] = b
                if c.temporal_mlp:       # image-family time branch (image/modeling_image.py:83-84,129-134); the video file removed it
                    for key, name, shape in (("tfc1", "temporal_mlp.fc1.weight", (f, d)), ("tfc2", "temporal_mlp.fc2.weight", (d, f))):
                        b = Block([(f"{p}.{name}", shape)], "mat"); blocks.append(b)
                        self._mat_blocks[f"{p}.{key}"] = b
                    vec += [(f"{p}.temporal_mlp.fc1.bias", (f,)), (f"{p}.temporal_mlp.fc2.bias", (d,)),
                            (f"{p}.temporal_layer_norm2.weight", (d,)), (f"{p}.temporal_layer_norm2.bias", (d,))]
                vec += [(f"{p}.temporal_embedding", (1, c.num_frames, d))]
                vec += [(f"{p}.temporal_attn.{n}_proj.bias", (d,)) for n in "qkv"]
                vec += [(f"{p}.temporal_attn.out_proj.bias", (d,)), (f"{p}.temporal_layer_norm1.weight", (d,)),
                        (f"{p}.temporal_layer_norm1.bias", (d,))]
            b = Block([(f"{p}.self_attn.{n}_proj.weight", (d, d)) for n in "qkv"], "mat"); blocks.append(b)
            self._mat_blocks[f"{p}.qkv"] = b
            for key, name, shape in (("out", "self_attn.out_proj.weight", (d, d)), ("fc1", "mlp.fc1.weight", (f, d)),
                                     ("fc2", "mlp.fc2.weight", (d, f))):
                b = Block([(f"{p}.{name}", shape)], "mat"); blocks.append(b)
                self._mat_blocks[f"{p}.{key}"] = b
            vec += [(f"{p}.self_attn.{n}_proj.bias", (d,)) for n in "qkv"]
            vec += [(f"{p}.self_attn.out_proj.bias", (d,)), (f"{p}.layer_norm1.weight", (d,)), (f"{p}.layer_norm1.bias", (d,)),
                    (f"{p}.mlp.fc1.bias", (f,)), (f"{p}.mlp.fc2.bias", (d,)), (f"{p}.layer_norm2.weight", (d,)),
                    (f"{p}.layer_norm2.bias", (d,))]
            blocks.append(Block(vec, "vec"))
        tail = "post_layernorm" if c.kind == "vision" else "final_layer_norm"
        tail_block = Block([(f"{tail}.weight", (d,)), (f"{tail}.bias", (d,))], "vec")
        blocks.append(tail_block)
        # LoRA sites: (mat-block key, first row inside the block's [rows, k_in] matrix, n_out, k_in, internal stem of the linear).
        # Targets (:778-783): q/k/v/out_proj of self_attn - or, with add_time_attn, of temporal_attn plus temporal_mlp.fc1 / fc2
        # (the video file names the MLP it removed; peft skips names that match nothing).
        self._lora_sites = []
        if self.lora:
            for i in range(c.num_hidden_layers):
                p = f"encoder.layers.{i}"
                attn, qkv_key, out_key = ("temporal_attn", "tqkv", "tout") if c.add_time_attn else ("self_attn", "qkv", "out")
                for j, n in enumerate("qkv"):
                    self._lora_sites.append((f"{p}.{qkv_key}", j * d, d, d, f"{p}.{attn}.{n}_proj"))
                self._lora_sites.append((f"{p}.{out_key}", 0, d, d, f"{p}.{attn}.out_proj"))
                if c.add_time_attn and c.temporal_mlp:
                    self._lora_sites.append((f"{p}.tfc1", 0, f, d, f"{p}.temporal_mlp.fc1"))
                    self._lora_sites.append((f"{p}.tfc2", 0, d, f, f"{p}.temporal_mlp.fc2"))
            for _, _, n_out, k_in, stem in self._lora_sites:     # one 256-byte aligned block per adapter pair, behind the tail LayerNorm
                blocks.append(Block([(f"{stem}.lora_A.default.weight", (c.lora_r, k_in)), (f"{stem}.lora_B.default.weight", (n_out, c.lora_r))], "vec"))
        if patch_block is not None:
            blocks.append(patch_block)
            self._mat_blocks["patch"] = patch_block
        self._store = FlatStore(blocks)
        self._lora_stems = {site[4] for site in self._lora_sites}
        self._merged = None
        if self.lora:
            # trainable flat ranges: [patch matrix | embeddings + pre_layrnorm] and [post_layernorm | adapters]; between them lie the
            # frozen encoder vectors, in front of them the frozen encoder matrices (whose gradient slots are scratch for dW)
            first_layer_vec = next(b for b in self._store.blocks if b.kind == "vec" and b.items[0][0].startswith("encoder.layers."))
            self._trainable = [(patch_block.offset, first_layer_vec.offset), (tail_block.offset, self._store.total)]
        # gradient buckets (engine.TrainEngine, N > 1 GPUs): flat ranges that become final as the backward walks the layers
        # 11 -> 0; `bucket_layers` consecutive layers per message, then [patch embedding | all vector parameters] at the end
        self.bucket_layers = 3
        self._bucket_hook = None
        self._param_names = list(self._store.index.keys())
        self._pub = {n: self._public_name(n) for n in self._param_names}     # internal (flat-store) name -> state-dict key
        for name in self._param_names:
            attach(self, self._pub[name], nn.Parameter(self._store.view(name), requires_grad=self.is_trainable(name)))
        n_pos = c.seq_len
        self.embeddings.register_buffer("position_ids", torch.arange(n_pos).expand((1, -1)).clone(), persistent=False)
        self._anchor = torch.zeros((), requires_grad=True)
        self._plist = [self._param(n) for n in self._param_names]
        self._shadow = {}
        self._shadow_version = None
        # load_state_dict writes through the parameters: belt and braces next to the version key below
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.mark_dirty())
        self._lp = None
        self._grad_fresh = False     # True between a backward of this tower and the optimizer step that consumes it
        self._accumulate = False     # set per backward: add to (instead of overwrite) the parameter gradients
        self._post_backward = None
        self._grad_cache = None   # set by engine.TrainEngine: called once the last backward kernel is enqueued
        self.reset_parameters(0 if seed is None else seed)

    # ------------------------------------------------------------------ parameters
    def named_flat(self):
        return self._store

    def _public_name(self, name: str) -> str:
        """state-dict key of an internal parameter name.  Plain towers: the same.  LoRA towers carry peft's wrapper names
        (image/modeling_image.py:793 replaces vision_model.encoder by get_peft_model(...)): ``encoder.layers.N.x`` becomes
        ``encoder.base_model.model.layers.N.x``, and a wrapped linear's own weight / bias move under ``.base_layer``."""
        if not self.lora or not name.startswith("encoder.layers."):
            return name
        stem, leaf = name.rsplit(".", 1)
        rest = name[len("encoder."):]
        if stem in self._lora_stems:
            rest = stem[len("encoder."):] + ".base_layer." + leaf
        return "encoder.base_model.model." + rest

    def is_trainable(self, name: str) -> bool:
        """LoRA towers: inside the encoder only the adapters train (peft freezes everything else it wraps)"""
        return not self.lora or not name.startswith("encoder.layers.") or ".lora_" in name

    def _param(self, name: str) -> nn.Parameter:
        return self.get_parameter(self._pub[name])

    def trainable_ranges(self):
        """flat ranges the optimizer and the all-reduce touch: everything, or (LoRA) the two ranges around the frozen encoder"""
        return list(self._trainable) if self.lora else [(0, self._store.total)]

    @torch.no_grad()
    def reset_parameters(self, seed: int):
        """Seeded init with the std's of CLIPPreTrainedModel._init_weights (image/modeling_image.py:179-230)."""
        c = self.config
        g = torch.Generator().manual_seed(seed)
        d, L = c.hidden_size, c.num_hidden_layers
        in_std, out_std, fc_std = d ** -0.5 * (2 * L) ** -0.5, d ** -0.5, (2 * d) ** -0.5
        cpu = {}
        for name in self._param_names:
            shape = self._store.index[name][1]
            leaf = name.split(".")[-2] + "." + name.split(".")[-1] if "." in name else name
            if ".lora_A." in name:          # peft's default: kaiming_uniform(a = sqrt(5)) = U(-1/sqrt(in), 1/sqrt(in)); B = 0
                t = (torch.rand(shape, generator=g) * 2 - 1) * shape[1] ** -0.5
            elif ".lora_B." in name:
                t = torch.zeros(shape)
            elif name.endswith("norm.weight") or name.endswith("norm1.weight") or name.endswith("norm2.weight"):
                t = torch.ones(shape)
            elif name.endswith(".bias"):
                t = torch.zeros(shape)
            elif "class_embedding" in name or "temporal_embedding" in name:
                t = torch.randn(shape, generator=g) * d ** -0.5
            elif "embedding" in name:
                t = torch.randn(shape, generator=g) * 0.02
            elif leaf.startswith(("q_proj", "k_proj", "v_proj")) or leaf.startswith("fc2"):
                t = torch.randn(shape, generator=g) * in_std
            elif leaf.startswith("out_proj"):
                t = torch.randn(shape, generator=g) * out_std
            elif leaf.startswith("fc1"):
                t = torch.randn(shape, generator=g) * fc_std
            else:
                raise AssertionError(name)
            cpu[name] = t
        for name, t in cpu.items():
            self._store.view(name).copy_(t)
        self.mark_dirty()

    def _rebind(self):
        for name in self._param_names:
            p = self._param(name)
            p.data = self._store.view(name)
            p.grad = None
        self._plist = [self._param(n) for n in self._param_names]
        self._merged = None
        self._grad_cache = None
        self._grad_fresh = False
        self._shadow.clear()
        self._shadow_version = None
        self._lp = None

    def _apply(self, fn, recurse=True):
        self._store.move(fn)
        self._rebind()
        for m in self.modules():
            for k, b in m._buffers.items():
                if b is not None:
                    m._buffers[k] = fn(b)
        self._anchor = torch.zeros((), requires_grad=True, device=self._store.master.device)
        return self

    def set_compute_dtype(self, dtype: torch.dtype):
        assert dtype in (torch.float32, torch.bfloat16)
        self.compute_dtype = dtype
        self._shadow.clear()
        self._shadow_version = None
        self._lp = None
        return self

    def mark_dirty(self):
        self._shadow_version = None

    def _version_key(self) -> int:
        """Changes whenever the fp32 master weights are written through torch: in place on the flat buffer, or through any
        parameter view.  ``p.data = view`` (what ``_rebind`` does on a device move, keeping the Parameter objects that
        optimizers already hold) gives every Parameter its OWN version counter, so the master's counter alone misses
        ``load_state_dict`` / ``torch.optim`` steps / ``p.mul_()`` after a ``.cuda()``: sum them all (~200 reads, a few us)."""
        v = self._store.master._version
        for p in self._plist:
            v += p._version
        return v

    def flat_master(self) -> torch.Tensor:
        return self._store.master

    def flat_grad(self) -> torch.Tensor:
        return self._store.ensure_grad()

    def _refresh_shadows(self):
        """(re)build the compute-dtype copies W [N,K] and W^T [K,N] of every GEMM weight block - one launch per tower"""
        st = self._store
        T = self.compute_dtype
        base = st.master
        if self.lora:
            # the GEMMs read W' = W + (alpha / r) B A (csrc/lora.hip): a second fp32 image of the matrix range, rebuilt from the frozen
            # base weights and the current adapters whenever either changed (once per optimizer step)
            if self._merged is None or self._merged.device != st.master.device:
                self._merged = torch.empty(st.vec_start, device=st.master.device, dtype=torch.float32)
                self._shadow.clear()
            self._merged.copy_(st.master[:st.vec_start])
            scale = float(self.config.lora_alpha) / float(self.config.lora_r)
            for key, row0, n_out, k_in, stem in self._lora_sites:
                b = self._mat_blocks[key]
                w = self._merged[b.offset:b.offset + b.numel].view(-1, k_in)[row0:row0 + n_out]
                ops.lora_merge(w, st.view(f"{stem}.lora_A.default.weight"), st.view(f"{stem}.lora_B.default.weight"), scale)
            base = self._merged
        if not self._shadow:
            entries = []
            self._cast_tile_range = []          # (flat offset of the block, first tile, tile count) in table order
            for key, b in self._mat_blocks.items():
                n_out = sum(it[1][0] for it in b.items)
                k_in = b.numel // n_out
                src = st.block_view(b, base).view(n_out, k_in)
                w = src if T == torch.float32 else torch.empty(n_out, k_in, device=src.device, dtype=T)
                wt = torch.empty(k_in, n_out, device=src.device, dtype=T)
                self._shadow[key] = (w, wt)
                ntile = ((n_out + 63) // 64) * ((k_in + 63) // 64)
                first = self._cast_tile_range[-1][1] + self._cast_tile_range[-1][2] if self._cast_tile_range else 0
                self._cast_tile_range.append((b.offset, first, ntile))
                entries.append((src, None if T == torch.float32 else w, wt))
            self._cast_table, self._cast_tiles = ops.build_cast_table(entries, st.master.device)
        ops.cast_weights_batched(self._cast_table, self._cast_tiles, ops.F32 if T == torch.float32 else ops.BF16)
        self._shadow_version = self._version_key()

    # ---- optimizer step fused with the shadow refresh (engine.TrainEngine) ----------------------------------------
    def vec_start(self) -> int:
        return self._store.vec_start

    def fused_update_ready(self) -> bool:
        """shadows exist and are current: the weight matrices may be updated through the cast-tile table"""
        st = self._store
        return not self.lora and bool(self._shadow) and st.master.is_cuda and self._shadow_version == self._version_key()

    def adam_and_refresh(self, grad, m, v, step, lr, beta1, beta2, eps, weight_decay, grad_scale=1.0, flat_range=None):
        """``flat_range`` = (lo, hi): only the weight matrices whose blocks start inside that range of the flat buffer (one
        gradient bucket, `layer_mat_range` / `tail_range`)"""
        st = self._store
        T = self.compute_dtype
        first, count = 0, self._cast_tiles
        if flat_range is not None:
            sel = [(f, n) for off, f, n in self._cast_tile_range if flat_range[0] <= off < flat_range[1]]
            first, count = (sel[0][0], sum(n for _, n in sel)) if sel else (0, 0)
            if sel and sel[-1][0] + sel[-1][1] - sel[0][0] != count:
                raise RuntimeError("adam_and_refresh: the blocks of a flat range are not contiguous in the tile table")
        ops.adam_cast_batched(self._cast_table, count, st.master, grad, m, v, step, lr, beta1, beta2, eps, weight_decay,
                              grad_scale, ops.F32 if T == torch.float32 else ops.BF16, tile_start=first)
        # (the kernels write through raw pointers: the master's version counter, hence the shadows' validity, is unchanged)

    def _ensure_ready(self):
        st = self._store
        if not st.master.is_cuda:
            raise _lib.MissmError("ClipTower runs only on an MI355X (move the module to cuda); there is no CPU fallback")
        if not self._shadow or self._shadow_version != self._version_key():
            self._refresh_shadows()
        if self._lp is None:
            self._build_layer_params()

    def _build_layer_params(self):
        st, c = self._store, self.config
        d = c.hidden_size
        self._lp = []
        st.ensure_grad()
        for i in range(c.num_hidden_layers):
            p = f"encoder.layers.{i}"
            L = SimpleNamespace()

            def vec3(prefix, buf):
                o = st.index[f"{p}.{prefix}.q_proj.bias"][0]
                return buf[o:o + 3 * d]

            L.qkv_b, L.g_qkv_b = vec3("self_attn", st.master), vec3("self_attn", st.grad)
            for nm, key in (("out_b", "self_attn.out_proj.bias"), ("ln1_w", "layer_norm1.weight"), ("ln1_b", "layer_norm1.bias"),
                            ("fc1_b", "mlp.fc1.bias"), ("fc2_b", "mlp.fc2.bias"), ("ln2_w", "layer_norm2.weight"),
                            ("ln2_b", "layer_norm2.bias")):
                setattr(L, nm, st.view(f"{p}.{key}")); setattr(L, "g_" + nm, st.gview(f"{p}.{key}"))
            for key in ("qkv", "out", "fc1", "fc2"):
                b = self._mat_blocks[f"{p}.{key}"]
                n_out = sum(it[1][0] for it in b.items)
                setattr(L, "g_" + key + "_w", st.block_view(b, st.grad).view(n_out, b.numel // n_out))
            if c.add_time_attn:
                L.tqkv_b, L.g_tqkv_b = vec3("temporal_attn", st.master), vec3("temporal_attn", st.grad)
                for nm, key in (("tout_b", "temporal_attn.out_proj.bias"), ("tln_w", "temporal_layer_norm1.weight"),
                                ("tln_b", "temporal_layer_norm1.bias")):
                    setattr(L, nm, st.view(f"{p}.{key}")); setattr(L, "g_" + nm, st.gview(f"{p}.{key}"))
                L.temb = st.view(f"{p}.temporal_embedding").view(c.num_frames, d)
                L.g_temb = st.gview(f"{p}.temporal_embedding").view(c.num_frames, d)
                if c.temporal_mlp:
                    for nm, key in (("tfc1_b", "temporal_mlp.fc1.bias"), ("tfc2_b", "temporal_mlp.fc2.bias"),
                                    ("tln2_w", "temporal_layer_norm2.weight"), ("tln2_b", "temporal_layer_norm2.bias")):
                        setattr(L, nm, st.view(f"{p}.{key}")); setattr(L, "g_" + nm, st.gview(f"{p}.{key}"))
                for key in ("tqkv", "tout") + (("tfc1", "tfc2") if c.temporal_mlp else ()):
                    b = self._mat_blocks[f"{p}.{key}"]
                    n_out = sum(it[1][0] for it in b.items)
                    setattr(L, "g_" + key + "_w", st.block_view(b, st.grad).view(n_out, b.numel // n_out))
            if self.lora:
                # frozen encoder vectors get no gradient (None pointers switch the kernels' parameter-gradient paths off); the
                # adapters of this layer's wrapped linears, per mat-block key: (first row, n_out, A, B, dA, dB)
                mats = {"g_qkv_w", "g_out_w", "g_fc1_w", "g_fc2_w", "g_tqkv_w", "g_tout_w", "g_tfc1_w", "g_tfc2_w"}
                for nm in [k for k in vars(L) if k.startswith("g_") and k not in mats]:
                    setattr(L, nm, None)
                L.lora = {}
                for key, row0, n_out, k_in, stem in self._lora_sites:
                    if key.startswith(p + "."):
                        a, bn = f"{stem}.lora_A.default.weight", f"{stem}.lora_B.default.weight"
                        L.lora.setdefault(key[len(p) + 1:], []).append((row0, n_out, st.view(a), st.view(bn), st.gview(a), st.gview(bn)))
            self._lp.append(L)

    def _w(self, key):
        return self._shadow[key]

    def layer_mat_range(self, lo_layer: int, hi_layer: int):
        """[start, end) of the flat buffer covering the weight matrices of layers lo_layer..hi_layer (inclusive)"""
        first = "tqkv" if self.config.add_time_attn else "qkv"
        a = self._mat_blocks[f"encoder.layers.{lo_layer}.{first}"]
        if hi_layer + 1 < self.config.num_hidden_layers:
            end = self._mat_blocks[f"encoder.layers.{hi_layer + 1}.{first}"].offset
        else:
            end = self._mat_blocks["patch"].offset if "patch" in self._mat_blocks else self._store.vec_start
        return a.offset, end

    def bucket_ranges(self):
        """the flat ranges in the order the backward hands them out (layers top-down in groups of `bucket_layers`, then the
        tail); together they cover the gradient buffer exactly once"""
        L, k = self.config.num_hidden_layers, self.bucket_layers
        out = [self.layer_mat_range(i, min(i + k, L) - 1) for i in reversed(range(L)) if i % k == 0]
        return out + [self.tail_range()]

    def tail_range(self):
        """[start, end): patch-embedding matrix (vision) + every vector parameter - final only when the backward has ended"""
        start = self._mat_blocks["patch"].offset if "patch" in self._mat_blocks else self._store.vec_start
        return start, self._store.total

    # ------------------------------------------------------------------ public forward (reference signature)
    def forward(self, pixel_values=None, output_attentions=None, output_hidden_states=None, return_dict=None, *,
                input_ids=None, attention_mask=None, position_ids=None, patch_keep=None):
        c = self.config
        if c.kind == "text" and input_ids is None and pixel_values is not None and pixel_values.dtype in (torch.int64, torch.int32):
            input_ids, pixel_values = pixel_values, None  # positional call of the text tower
        if c.kind == "vision":
            if pixel_values is None:
                raise ValueError("You have to specify pixel_values")
            inputs = (pixel_values, patch_keep)
        else:
            if input_ids is None:
                raise ValueError("You have to specify input_ids")
            inputs = (input_ids, attention_mask)
        self._ensure_ready()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        params = tuple(self._param(n) for n in self._param_names if self.is_trainable(n)) if (need_grad and _GRAD_MODE == "autograd") else ()
        last, pooled = _TowerFn.apply(self._anchor, self, inputs, need_grad, *params)
        return (last, pooled)

    # ------------------------------------------------------------------ forward / backward implementation
    def _forward_impl(self, inputs, save: bool):
        return forward_lanes([self], [inputs], save)[0]

    def _backward_impl(self, s, d_last, d_pooled):
        backward_lanes([self], [s], [d_last], [d_pooled])

    def _begin_backward(self):
        """torch.autograd semantics for .grad: a backward ADDS to gradients that have not been consumed yet - a second
        micro-batch, the reference's teacher and student sharing one encoder, a tower called twice in a step.  "Consumed"
        = the engine's optimizer step ran (_grad_fresh cleared) or zero_grad(set_to_none=True) detached the views; after
        zero_grad(set_to_none=False) the buffers hold zeros, and adding to them is the same as storing.  In "autograd"
        gradient mode the values are handed to autograd, which does its own accumulation."""
        st = self._store
        first = next(n for n in self._param_names if self.is_trainable(n))
        fp = self._param(first)
        self._accumulate = bool(self._grad_fresh and _GRAD_MODE == "direct" and fp.grad is not None
                                and st.grad is not None and fp.grad.data_ptr() == st.gview(first).data_ptr())
        if not self._accumulate:
            st.zero_accumulated()

    def attach_grads(self):
        """p.grad of every parameter = its view into the flat gradient buffer.  Runs after every backward on the host thread
        that feeds the GPU, so the (parameter, view) pairs are resolved once: walking ~200 dotted names per call cost 0.7 ms,
        during which the stream of the following tower's backward sat empty."""
        st = self._store
        grad = st.ensure_grad()
        cache = self._grad_cache
        if cache is None or cache[0] is not grad:
            cache = self._grad_cache = (grad, [(self._param(n), st.gview(n)) for n in self._param_names if self.is_trainable(n)])
        for p, gv in cache[1]:
            if p.grad is not gv:
                p.grad = gv



# ======================================================================================================================
# Lock-step execution of `lanes` = one or several shape-identical towers (same config, compute dtype and input geometry).
# Every op that is not a GEMM is launched once per lane; every linear, its input gradient and its weight gradient are ONE
# grouped call (ops.gemm_grouped): the library then shares one tile grid between the lanes, so that four B x 197-row towers
# fill the chip like one long tower (their own 128x128 grids fill 59-88 % of it and run at ~480 TFLOP/s against ~1000 for the
# video tower's).  With one lane this is exactly the single-tower path.
# ======================================================================================================================
_SPARSE_LAST = os.environ.get("MISSM_SPARSE_LAST", "1") != "0"   # 0: the last layer's backward runs on all rows (A/B, debugging)

import contextlib
import threading
_POOLED = threading.local()


@contextlib.contextmanager
def pooled_output_only():
    """Inside this context a vision tower's forward may leave `last_hidden_state` uncomputed (it is returned as None): the bundle
    reads the pooled output only (`self.modality_encoder[key](**value)[1]`, languagebind/__init__.py:78), and the pooled output is
    post_layernorm(h[:, 0]) - of the LAST layer's out-projection and MLP only the CLS rows are needed.  A tower called directly
    (ClipTower.forward) always returns the full last_hidden_state."""
    prev = getattr(_POOLED, "on", False)
    _POOLED.on = True
    try:
        yield
    finally:
        _POOLED.on = prev


def _linear_bwd_lanes(lanes, dys, xs, wts, g_ws, g_bs, rows, dx_outs=None, act=ops.ACT_NONE, aux_ins=None, lora_sites=None):
    """dW = dy^T x: TN GEMM reading dy / x where they lie (split-K slices summed by the library's reduce kernel, result STORED
    into the fp32 gradient - or ADDED to it when this backward accumulates, ClipTower._begin_backward); db += colsum(dy) riding
    in the same GEMM (tail of the gradient buffer); dx = dy W through the transposed compute-dtype shadow (NT form).
    LoRA towers (`lora_sites` = per lane None: a frozen linear, or its adapters): a frozen linear has no weight gradient at all; a
    wrapped one writes the full dW into its (scratch) gradient slot and derives dA / dB from it (csrc/lora.hip), which accumulate."""
    if lora_sites is None:
        ops.gemm_grouped(dys, xs, g_ws, trans_a=True, trans_b=True, splitk=0, K=rows, colsum_a=g_bs, accumulate=lanes[0]._accumulate)
    elif lora_sites[0] is not None:
        ops.gemm_grouped(dys, xs, g_ws, trans_a=True, trans_b=True, splitk=0, K=rows)
        for t, g_w, sites in zip(lanes, g_ws, lora_sites):
            scale = float(t.config.lora_alpha) / float(t.config.lora_r)
            for row0, n_out, a, b, da, db in sites:
                ops.lora_grad(g_w[row0:row0 + n_out], a, b, da, db, scale)
    if dx_outs is not None:
        ops.gemm_grouped(dys, wts, dx_outs, act=act, aux_in=aux_ins, M=rows)
    return dx_outs


def lanes_compatible(towers, inputs) -> bool:
    t0 = towers[0]
    if len(towers) == 1:
        return True
    if any(t.config != t0.config or t.compute_dtype != t0.compute_dtype for t in towers):
        return False
    if t0.config.kind != "vision":
        return False
    shapes = {tuple(i[0].shape) for i in inputs}
    return len(shapes) == 1 and all(i[0].device == inputs[0][0].device for i in inputs)


def forward_lanes(towers, inputs, save: bool):
    """returns [(state, last_hidden_state, pooled_output)] per lane"""
    t0 = towers[0]
    c, T = t0.config, t0.compute_dtype
    G = range(len(towers))
    sts = [t._store for t in towers]
    dev = sts[0].master.device
    d, f, H = c.hidden_size, c.intermediate_size, c.num_attention_heads
    hd = d // H
    S = c.seq_len
    ss = [SimpleNamespace(layers=[], save=save) for _ in G]
    f32 = dict(device=dev, dtype=torch.float32)
    E = lambda *shape, **kw: [torch.empty(*shape, **kw) for _ in G]      # noqa: E731  one buffer per lane

    def EA(n0, *rest, **kw):
        """one allocation, one slice per lane (lanes that share a launch over their leading dimension)"""
        whole = torch.empty(n0 * len(towers), *rest, **kw)
        return whole, [whole[g * n0:(g + 1) * n0] for g in G]

    if c.kind == "vision":
        pxs = []
        for g in G:
            px = inputs[g][0]
            if px.dim() == 7:
                b_new, pair_new, Tf, bs_new, ch, hh, ww = px.shape
                B = b_new * pair_new * bs_new
                px = px.reshape(B * Tf, ch, hh, ww)
            elif px.dim() == 5:
                B, _, Tf, _, _ = px.shape
            elif px.dim() == 4:
                B, Tf = px.shape[0], 1
            else:
                raise ValueError(f"pixel_values must be 4-D, 5-D or 7-D, got {px.dim()}-D")
            px = px.to(device=dev, dtype=torch.float32)
            if px.stride(-1) != 1 or px.stride(-2) != px.shape[-1]:
                px = px.contiguous()
            if px.shape[-1] != c.image_hw[1] or px.shape[-2] != c.image_hw[0] or px.shape[1] != c.num_channels:
                raise ValueError(f"pixel_values spatial/channel shape {tuple(px.shape)} does not match the tower config")
            if c.add_time_attn and Tf != c.num_frames:
                raise ValueError(f"time attention is configured for {c.num_frames} frames, got {Tf}")
            pxs.append(px)
        N = B * Tf
        P = S - 1
        Kp = (c.num_channels * c.patch_size ** 2 + 7) // 8 * 8      # (padded like the patch matrix's rows)
        U = E(N * P, Kp, device=dev, dtype=T)
        for g in G:
            ops.unfold_patches(pxs[g], U[g], c.patch_size)
        pe = E(N * P, d, device=dev, dtype=T)
        ops.gemm_grouped(U, [t._w("patch")[0] for t in towers], pe)
        # PatchDropout (image/modeling_image.py:30-63,647): training mode and prob > 0 - the class token and K kept patch tokens
        # of every frame go on; the gather rides in pre_layrnorm's row addressing (row r reads embedding row r + off[r]), the
        # rest of the tower simply runs on S = 1 + K tokens per frame
        S0, offs = S, [None for _ in G]
        drop = [bool(c.force_patch_dropout > 0 and t.training) for t in towers]
        if any(drop):
            if not all(drop):
                raise RuntimeError("towers run in lock-step must agree on train / eval mode (PatchDropout is active in training only)")
            K = t0.patch_dropout.num_keep(P)
            S = 1 + K
            for g in G:
                idx = inputs[g][1] if len(inputs[g]) > 1 and inputs[g][1] is not None else towers[g].patch_dropout.keep_indices(B, Tf, P)
                idx = torch.as_tensor(idx).to("cpu", torch.int64)
                want = (N if Tf == 1 else B, K)
                if tuple(idx.shape) != want or int(idx.min()) < 0 or int(idx.max()) >= P:
                    raise ValueError(f"patch_keep must hold {want} indices into the {P} patch tokens")
                if Tf != 1:
                    idx = idx.unsqueeze(1).repeat(1, Tf, 1).reshape(N, K)          # 'b t n -> (b t) n' (:55-56)
                src = torch.cat([torch.zeros(N, 1, dtype=torch.int64), 1 + idx], dim=1) + torch.arange(N)[:, None] * S0
                offs[g] = (src.reshape(-1) - torch.arange(N * S)).to(torch.int32).to(dev)
        x0, h = E(N * S0, d, **f32), E(N * S, d, **f32)
        m0, r0 = E(N * S, **f32), E(N * S, **f32)
        for g in G:
            st = sts[g]
            ops.embed_assemble(pe[g], st.view("embeddings.class_embedding"), st.view("embeddings.position_embedding.weight"), x0[g], N, S0, d)
            ops.layernorm_fwd(x0[g], st.view("pre_layrnorm.weight"), st.view("pre_layrnorm.bias"), h[g], m0[g], r0[g], N * S, d, c.layer_norm_eps,
                              in_off=offs[g])
            ss[g].emb = (U[g], x0[g], m0[g], r0[g], offs[g], S0) if save else None
        causal, key_mask = False, [None for _ in G]
    else:
        (ids, amask), = inputs          # (text towers are never grouped)
        ids = ids.to(dev).view(-1, ids.shape[-1]).contiguous().long()
        B, Sx = ids.shape
        if Sx > c.max_position_embeddings:
            raise ValueError("sequence longer than max_position_embeddings")
        S, Tf, N = Sx, 1, B
        h = E(N * S, d, **f32)
        ops.token_embed_fwd(ids, sts[0].view("embeddings.token_embedding.weight"), sts[0].view("embeddings.position_embedding.weight"), h[0], B, S, d)
        causal = True
        key_mask = [amask.to(dev).view(B, S).to(torch.int32).contiguous() if amask is not None else None]
        ss[0].ids = ids
    rows = N * S
    for g in G:
        ss[g].geom = (B, Tf, N, S, rows)
    act = ops.ACT_CODE[c.hidden_act]
    W = lambda key, which: [t._w(key)[which] for t in towers]            # noqa: E731
    pooled_only = (getattr(_POOLED, "on", False) and _SPARSE_LAST and c.kind == "vision" and not t0.lora and c.num_hidden_layers > 0)
    for g in G:
        ss[g].pooled_only = pooled_only
    for i in range(c.num_hidden_layers):
        L = [t._lp[i] for t in towers]
        pfx = f"encoder.layers.{i}"
        recs = [SimpleNamespace() for _ in G]
        if c.add_time_attn:
            xt = E(rows, d, device=dev, dtype=T)
            mt, rt = E(rows, **f32), E(rows, **f32)
            for g in G:
                ops.layernorm_fwd(h[g], L[g].tln_w, L[g].tln_b, xt[g], mt[g], rt[g], rows, d, c.layer_norm_eps,
                                  add=L[g].temb if Tf != 1 else None, add_div=S, add_mod=Tf)
            qkv = E(rows, 3 * d, device=dev, dtype=T)
            ops.gemm_grouped(xt, W(f"{pfx}.tqkv", 0), qkv, bias=[l.tqkv_b for l in L])
            ctx = E(rows, d, device=dev, dtype=T)
            lse = E(B * S * H * Tf, **f32)
            for g in G:
                ops.attention_fwd(qkv[g], ctx[g], lse[g], B * S, Tf, H, hd, seq_div=S, seq_outer=Tf * S, seq_inner=1, tok_stride=S)
            h2 = E(rows, d, **f32)
            ops.gemm_grouped(ctx, W(f"{pfx}.tout", 0), h2, bias=[l.tout_b for l in L], resid=h)
            if save:
                for g in G:
                    recs[g].t = (h[g], xt[g], mt[g], rt[g], qkv[g], ctx[g], lse[g])
            h = h2
            if c.temporal_mlp:
                # h = h + temporal_mlp(temporal_layer_norm2(h)) (image/modeling_image.py:129-134): LayerNorm and MLP act row by row,
                # so the '(b t) n d <-> (b n) t d' rearrangements around them change nothing
                xm = E(rows, d, device=dev, dtype=T)
                mm, rm = E(rows, **f32), E(rows, **f32)
                ops.layernorm_fwd_lanes(h, [l.tln2_w for l in L], [l.tln2_b for l in L], xm, mm, rm, rows, d, c.layer_norm_eps)
                at = E(rows, f, device=dev, dtype=T)
                ut = E(rows, f, device=dev, dtype=T) if save else None
                ops.gemm_grouped(xm, W(f"{pfx}.tfc1", 0), at, bias=[l.tfc1_b for l in L], act=act, aux_out=ut)
                h3 = E(rows, d, **f32)
                ops.gemm_grouped(at, W(f"{pfx}.tfc2", 0), h3, bias=[l.tfc2_b for l in L], resid=h)
                if save:
                    for g in G:
                        recs[g].tm = (h[g], xm[g], mm[g], rm[g], ut[g], at[g])
                h = h3
        x1 = E(rows, d, device=dev, dtype=T)
        m1, r1 = E(rows, **f32), E(rows, **f32)
        ops.layernorm_fwd_lanes(h, [l.ln1_w for l in L], [l.ln1_b for l in L], x1, m1, r1, rows, d, c.layer_norm_eps)
        # (the lanes' QKV / context / log-sum-exp buffers are slices of ONE allocation: the sequences of all lanes are one
        #  attention launch - four 384-(frame, head) launches fill 512 workgroup slots to 75 % each, one 1536-unit launch runs
        #  three full rounds in which one workgroup's K / V fetch overlaps another's arithmetic)
        qkv_all, qkv = EA(rows, 3 * d, device=dev, dtype=T)
        ops.gemm_grouped(x1, W(f"{pfx}.qkv", 0), qkv, bias=[l.qkv_b for l in L])
        ctx_all, ctx = EA(rows, d, device=dev, dtype=T)
        lse_all, lse = EA(N * H * S, **f32)
        if len(towers) > 1:
            ops.attention_fwd(qkv_all, ctx_all, lse_all, N * len(towers), S, H, hd, causal=causal)
        else:
            ops.attention_fwd(qkv[0], ctx[0], lse[0], N, S, H, hd, causal=causal, key_mask=key_mask[0])
        cls_fwd = pooled_only and i == c.num_hidden_layers - 1
        if cls_fwd:
            # last layer, pooled output only: out-projection, LayerNorm 2 and the MLP on the N = B * T CLS rows (rows n * S of the full
            # buffers through strided views; the LayerNorm / MLP operands compact [N, .]); the other rows of h2 / h3 stay unwritten
            cl = lambda t, w: t.view(N, S, w)[:, 0]          # noqa: E731
            h2 = E(rows, d, **f32)
            ops.gemm_grouped([cl(t, d) for t in ctx], W(f"{pfx}.out", 0), [cl(t, d) for t in h2], bias=[l.out_b for l in L], resid=[cl(t, d) for t in h])
            x2 = E(N, d, device=dev, dtype=T)
            m2, r2 = E(N, **f32), E(N, **f32)
            for g in G:
                ops.layernorm_fwd(h2[g], L[g].ln2_w, L[g].ln2_b, x2[g], m2[g], r2[g], N, d, c.layer_norm_eps, in_mul=S)
            a = E(N, f, device=dev, dtype=T)
            u = E(N, f, device=dev, dtype=T) if save else None
            ops.gemm_grouped(x2, W(f"{pfx}.fc1", 0), a, bias=[l.fc1_b for l in L], act=act, aux_out=u)
            h3 = E(rows, d, **f32)
            ops.gemm_grouped(a, W(f"{pfx}.fc2", 0), [cl(t, d) for t in h3], bias=[l.fc2_b for l in L], resid=[cl(t, d) for t in h2])
        else:
            h2 = E(rows, d, **f32)
            ops.gemm_grouped(ctx, W(f"{pfx}.out", 0), h2, bias=[l.out_b for l in L], resid=h)
            x2 = E(rows, d, device=dev, dtype=T)
            m2, r2 = E(rows, **f32), E(rows, **f32)
            ops.layernorm_fwd_lanes(h2, [l.ln2_w for l in L], [l.ln2_b for l in L], x2, m2, r2, rows, d, c.layer_norm_eps)
            a = E(rows, f, device=dev, dtype=T)
            u = E(rows, f, device=dev, dtype=T) if save else None
            ops.gemm_grouped(x2, W(f"{pfx}.fc1", 0), a, bias=[l.fc1_b for l in L], act=act, aux_out=u)
            h3 = E(rows, d, **f32)
            ops.gemm_grouped(a, W(f"{pfx}.fc2", 0), h3, bias=[l.fc2_b for l in L], resid=h2)
        if save:
            for g in G:
                recs[g].a = (h[g], x1[g], m1[g], r1[g], qkv[g], ctx[g], lse[g])
                recs[g].m = (h2[g], x2[g], m2[g], r2[g], u[g], a[g])
                recs[g].cls = cls_fwd            # the MLP block's saved operands are compact [N, .] (CLS rows)
                recs[g].a_all = (qkv_all, ctx_all, lse_all) if g == 0 else None
                ss[g].layers.append(recs[g])
        h = h3
    out = []
    for g in G:
        st, s = sts[g], ss[g]
        s.key_mask, s.causal = key_mask[g], causal
        # pooling (+ final LayerNorm)
        if c.kind == "vision":
            pl = torch.empty(N, d, **f32)
            mp, rp = torch.empty(N, **f32), torch.empty(N, **f32)
            ops.layernorm_fwd(h[g], st.view("post_layernorm.weight"), st.view("post_layernorm.bias"), pl, mp, rp, N, d,
                              c.layer_norm_eps, in_mul=S)
            if Tf > 1:
                pooled = torch.empty(B, d, **f32)
                ops.mean_rows(pl, pooled, B, Tf, d)
            else:
                pooled = pl
            last = None if pooled_only else h[g].view(N, S, d)      # (pooled only: the non-CLS rows of h were never computed)
            s.pool = (h[g], mp, rp, None, None, None) if save else None
        else:
            last = torch.empty(rows, d, **f32)
            mf, rf = torch.empty(rows, **f32), torch.empty(rows, **f32)
            gw, gb = st.view("final_layer_norm.weight"), st.view("final_layer_norm.bias")
            ops.layernorm_fwd(h[g], gw, gb, last, mf, rf, rows, d, c.layer_norm_eps)
            eot = torch.empty(B, dtype=torch.int32, device=dev)
            ops.argmax_rows(s.ids, eot, B, S)
            pooled = torch.empty(B, d, **f32)
            mp, rp = torch.empty(B, **f32), torch.empty(B, **f32)
            ops.layernorm_fwd(h[g], gw, gb, pooled, mp, rp, B, d, c.layer_norm_eps, in_mul=S, in_off=eot)
            last = last.view(B, S, d)
            s.pool = (h[g], mp, rp, eot, mf, rf) if save else None
        out.append((s, last, pooled))
    return out


def backward_lanes(towers, states, d_lasts, d_pooleds):
    t0 = towers[0]
    c, T = t0.config, t0.compute_dtype
    G = range(len(towers))
    sts = [t._store for t in towers]
    dev = sts[0].master.device
    d, f, H = c.hidden_size, c.intermediate_size, c.num_attention_heads
    hd = d // H
    B, Tf, N, S, rows = states[0].geom
    for t in towers:
        t._begin_backward()
    if any(t._accumulate != t0._accumulate for t in towers):
        raise RuntimeError("towers run in lock-step must agree on accumulating or overwriting their gradients")
    f32 = dict(device=dev, dtype=torch.float32)
    E = lambda *shape, **kw: [torch.empty(*shape, **kw) for _ in G]      # noqa: E731
    tail = "post_layernorm" if c.kind == "vision" else "final_layer_norm"
    dh, dh_T = [], E(rows, d, device=dev, dtype=T)
    for g in G:
        st, s, d_last, d_pooled = sts[g], states[g], d_lasts[g], d_pooleds[g]
        gv = st.gview
        h_fin, mp, rp, eot, mf, rf = s.pool
        if d_last is not None and c.kind == "vision":
            dhg = d_last.reshape(rows, d).to(torch.float32).clone()
        else:
            dhg = torch.zeros(rows, d, **f32)
        gw = st.view(f"{tail}.weight")
        if d_pooled is not None:
            dp = d_pooled.to(torch.float32).contiguous()
            ops.layernorm_bwd(dp, h_fin, mp, rp, gw, dhg, gv(f"{tail}.weight"), gv(f"{tail}.bias"), N if c.kind == "vision" else B, d,
                              accumulate=True, dy_div=Tf if c.kind == "vision" else 1,
                              dy_scale=1.0 / Tf if c.kind == "vision" else 1.0, in_mul=S, in_off=eot)
        if d_last is not None and c.kind == "text":
            dl = d_last.reshape(rows, d).to(torch.float32).contiguous()
            ops.layernorm_bwd(dl, h_fin, mf, rf, gw, dhg, gv(f"{tail}.weight"), gv(f"{tail}.bias"), rows, d, accumulate=True)
        ops.cast_rows(dhg, dh_T[g], rows, d)
        dh.append(dhg)
    dact = ops.ACT_GRAD[ops.ACT_CODE[c.hidden_act]]
    W = lambda key, which: [t._w(key)[which] for t in towers]            # noqa: E731
    # only the pooled output carries a gradient (every training loop of the reference: `modality_encoder[key](**value)[1]`,
    # languagebind/__init__.py:78): the residual gradient enters the last layer non-zero on the CLS rows only
    cls_only = (_SPARSE_LAST and c.kind == "vision" and not t0.lora and all(dl is None for dl in d_lasts) and c.num_hidden_layers > 0)
    if any(getattr(st_, "pooled_only", False) for st_ in states) and not cls_only:
        raise RuntimeError("a pooled-output-only forward cannot be differentiated through last_hidden_state")
    for i in reversed(range(c.num_hidden_layers)):
        L = [t._lp[i] for t in towers]
        # LoRA towers: per linear None (frozen: no dW) or its adapters (dW into scratch, then dA / dB); plain towers: None = the usual dW
        LS = (lambda key: [l.lora.get(key) for l in L]) if t0.lora else (lambda key: None)       # noqa: E731
        pfx = f"encoder.layers.{i}"
        recs = [states[g].layers[i] for g in G]
        h2, x2, m2, r2, u, a = (list(v) for v in zip(*[r.m for r in recs]))
        hin, x1, m1, r1, qkv, ctx, lse = (list(v) for v in zip(*[r.a for r in recs]))
        if cls_only and i == c.num_hidden_layers - 1:
            # The LAST layer's MLP block and out-projection see a gradient that is non-zero on the CLS rows only (the loss reads
            # post_layernorm(h[:, 0]) and nothing else; LayerNorm, the MLP and the linears act row by row): their input / weight
            # gradients are formed from the N = B * T CLS rows - strided views of the saved activations, no copies - instead of N * S
            # rows of which all the others are exactly zero.  Same sums, 1 / S of the work: seven full-size GEMMs and a LayerNorm
            # backward per tower and step (1.8 ms of kernel time at B = 32).  Behind the attention backward the gradient is dense.
            cl = lambda t, w: t.view(N, S, w)[:, 0]          # noqa: E731  rows n * S of a [rows, w] buffer, row pitch S * w
            compact = bool(getattr(recs[0], "cls", False))   # the forward ran this block on the CLS rows: its saved operands are [N, .]
            cm = (lambda t, w: t) if compact else cl         # noqa: E731
            cs = (lambda t: t) if compact else (lambda t: t.view(N, S)[:, 0].contiguous())     # noqa: E731  row statistics
            dh_c = [cl(dh_T[g], d) for g in G]
            du = E(N, f, device=dev, dtype=T)
            _linear_bwd_lanes(towers, dh_c, [cm(t, f) for t in a], W(f"{pfx}.fc2", 1), [l.g_fc2_w for l in L], [l.g_fc2_b for l in L], N,
                              dx_outs=du, act=dact, aux_ins=[cm(t, f) for t in u])
            dx2 = E(N, d, device=dev, dtype=T)
            _linear_bwd_lanes(towers, du, [cm(t, d) for t in x2], W(f"{pfx}.fc1", 1), [l.g_fc1_w for l in L], [l.g_fc1_b for l in L], N, dx_outs=dx2)
            for g in G:     # LayerNorm 2 backward on the gathered rows: x, the running gradient and its compute-dtype copy at rows n * S
                ops.layernorm_bwd(dx2[g], h2[g], cs(m2[g]), cs(r2[g]), L[g].ln2_w, dh[g],
                                  L[g].g_ln2_w, L[g].g_ln2_b, N, d, accumulate=True, in_mul=S, dx_cast=dh_T[g])
            dctx_all = torch.zeros(rows * len(towers), d, device=dev, dtype=T)
            dctx = [dctx_all[g * rows:(g + 1) * rows] for g in G]
            _linear_bwd_lanes(towers, dh_c, [cl(t, d) for t in ctx], W(f"{pfx}.out", 1), [l.g_out_w for l in L], [l.g_out_b for l in L], N,
                              dx_outs=[cl(t, d) for t in dctx])
        else:
            # ---- MLP block: h3 = h2 + fc2(act(fc1(LN2(h2))))
            du = E(rows, f, device=dev, dtype=T)
            _linear_bwd_lanes(towers, dh_T, a, W(f"{pfx}.fc2", 1), [l.g_fc2_w for l in L], [l.g_fc2_b for l in L], rows, dx_outs=du, act=dact, aux_ins=u, lora_sites=LS("fc2"))
            dx2 = E(rows, d, device=dev, dtype=T)
            _linear_bwd_lanes(towers, du, x2, W(f"{pfx}.fc1", 1), [l.g_fc1_w for l in L], [l.g_fc1_b for l in L], rows, dx_outs=dx2, lora_sites=LS("fc1"))
            ops.layernorm_bwd_lanes(dx2, h2, m2, r2, [l.ln2_w for l in L], dh, [l.g_ln2_w for l in L], [l.g_ln2_b for l in L], rows, d, dh_T)
            # ---- attention block: h2 = h + out(attn(qkv(LN1(h))))
            dctx_all = torch.empty(rows * len(towers), d, device=dev, dtype=T)
            dctx = [dctx_all[g * rows:(g + 1) * rows] for g in G]
            _linear_bwd_lanes(towers, dh_T, ctx, W(f"{pfx}.out", 1), [l.g_out_w for l in L], [l.g_out_b for l in L], rows, dx_outs=dctx, lora_sites=LS("out"))
        if len(towers) > 1:      # one launch over every lane's sequences (forward_lanes made the saved buffers slices of one allocation)
            qkv_all, ctx_all, lse_all = recs[0].a_all
            dqkv_all = torch.empty(rows * len(towers), 3 * d, device=dev, dtype=T)
            dqkv = [dqkv_all[g * rows:(g + 1) * rows] for g in G]
            ops.attention_bwd(qkv_all, ctx_all, dctx_all, lse_all, dqkv_all, N * len(towers), S, H, hd, causal=states[0].causal)
        else:
            dqkv = E(rows, 3 * d, device=dev, dtype=T)
            ops.attention_bwd(qkv[0], ctx[0], dctx[0], lse[0], dqkv[0], N, S, H, hd, causal=states[0].causal, key_mask=states[0].key_mask)
        dx1 = E(rows, d, device=dev, dtype=T)
        _linear_bwd_lanes(towers, dqkv, x1, W(f"{pfx}.qkv", 1), [l.g_qkv_w for l in L], [l.g_qkv_b for l in L], rows, dx_outs=dx1, lora_sites=LS("qkv"))
        ops.layernorm_bwd_lanes(dx1, hin, m1, r1, [l.ln1_w for l in L], dh, [l.g_ln1_w for l in L], [l.g_ln1_b for l in L], rows, d, dh_T)
        if c.temporal_mlp:
            hin, xm, mm, rm, ut, at = (list(v) for v in zip(*[r.tm for r in recs]))
            dut = E(rows, f, device=dev, dtype=T)
            _linear_bwd_lanes(towers, dh_T, at, W(f"{pfx}.tfc2", 1), [l.g_tfc2_w for l in L], [l.g_tfc2_b for l in L], rows, dx_outs=dut, act=dact, aux_ins=ut, lora_sites=LS("tfc2"))
            dxm = E(rows, d, device=dev, dtype=T)
            _linear_bwd_lanes(towers, dut, xm, W(f"{pfx}.tfc1", 1), [l.g_tfc1_w for l in L], [l.g_tfc1_b for l in L], rows, dx_outs=dxm, lora_sites=LS("tfc1"))
            ops.layernorm_bwd_lanes(dxm, hin, mm, rm, [l.tln2_w for l in L], dh, [l.g_tln2_w for l in L], [l.g_tln2_b for l in L], rows, d, dh_T)
        if c.add_time_attn:
            hin, xt, mt, rt, qkv, ctx, lse = (list(v) for v in zip(*[r.t for r in recs]))
            dctx = E(rows, d, device=dev, dtype=T)
            _linear_bwd_lanes(towers, dh_T, ctx, W(f"{pfx}.tout", 1), [l.g_tout_w for l in L], [l.g_tout_b for l in L], rows, dx_outs=dctx, lora_sites=LS("tout"))
            dqkv = E(rows, 3 * d, device=dev, dtype=T)
            for g in G:
                ops.attention_bwd(qkv[g], ctx[g], dctx[g], lse[g], dqkv[g], B * S, Tf, H, hd, seq_div=S, seq_outer=Tf * S, seq_inner=1, tok_stride=S)
            dxt = E(rows, d, device=dev, dtype=T)
            _linear_bwd_lanes(towers, dqkv, xt, W(f"{pfx}.tqkv", 1), [l.g_tqkv_w for l in L], [l.g_tqkv_b for l in L], rows, dx_outs=dxt, lora_sites=LS("tqkv"))
            for g in G:
                # d temporal_embedding[t] = sum over (b, n) of the updated residual gradient: group sums riding in the LayerNorm backward
                # (a separate column-sum pass over the 155 MB video gradient until round 3)
                temb = L[g].g_temb if Tf != 1 else None
                ops.layernorm_bwd(dxt[g], hin[g], mt[g], rt[g], L[g].tln_w, dh[g], L[g].g_tln_w, L[g].g_tln_b, rows, d, accumulate=True, dx_cast=dh_T[g],
                                  gsum=temb, gs_div=S, gs_mod=Tf)
        for g in G:
            states[g].layers[i] = None
        if i % t0.bucket_layers == 0:
            # the weight-matrix gradients of layers i .. i+bucket_layers-1 are final (their kernels are enqueued): hand the
            # range to the engine, whose all-reduce then travels while the layers below are still differentiating
            for t in towers:
                if t._bucket_hook is not None:
                    t._bucket_hook(t, *t.layer_mat_range(i, min(i + t.bucket_layers, c.num_hidden_layers) - 1))
    # ---- embeddings
    if c.kind == "vision":
        S0 = states[0].emb[5]               # tokens per frame BEFORE PatchDropout (== S when it was the identity)
        P = S0 - 1
        dpe = E(N * P, d, device=dev, dtype=T)
        Us = []
        for g in G:
            st, s = sts[g], states[g]
            gv = st.gview
            U, x0, m0, r0, off, _ = s.emb
            # (PatchDropout: the gradient is scattered back to the kept tokens' embedding rows, the dropped ones get zero)
            dx = torch.empty(rows, d, **f32) if off is None else torch.zeros(N * S0, d, **f32)
            if off is None:
                # position-embedding gradient = sum over frames of dx[n, s, :]: group sums (group = row % S) riding in this LayerNorm
                # backward instead of two column-sum passes over dx; class_embedding's gradient is row 0 of the same sums
                psum = torch.zeros(S0, d, **f32)
                ops.layernorm_bwd(dh[g], x0, m0, r0, st.view("pre_layrnorm.weight"), dx, gv("pre_layrnorm.weight"), gv("pre_layrnorm.bias"),
                                  rows, d, accumulate=False, gsum=psum, gs_div=0, gs_mod=S0)
                gv("embeddings.position_embedding.weight").view(S0, d).add_(psum)
                gv("embeddings.class_embedding").add_(psum[0])
            else:
                ops.layernorm_bwd(dh[g], x0, m0, r0, st.view("pre_layrnorm.weight"), dx, gv("pre_layrnorm.weight"), gv("pre_layrnorm.bias"),
                                  rows, d, accumulate=False, in_off=off)
                # position-embedding gradient: sum over frames of dx[n, s, :] = a plain column sum of dx viewed as [N, S0*d]
                ops.colsum(dx.view(N, S0 * d), gv("embeddings.position_embedding.weight").view(S0 * d), R=N)
                ops.colsum(dx.view(N, S0 * d)[:, :d], gv("embeddings.class_embedding"), R=N)
            ops.cast_rows(dx, dpe[g], N * P, d, rdiv=P, roff=1)
            Us.append(U)
        gwp = []
        for t in towers:
            b = t._mat_blocks["patch"]
            gwp.append(t._store.block_view(b, t._store.grad).view(d, b.numel // d))
        _linear_bwd_lanes(towers, dpe, Us, None, gwp, None, N * P)
    else:
        st, s = sts[0], states[0]
        ops.token_embed_bwd(s.ids, dh[0], st.gview("embeddings.token_embedding.weight"), st.gview("embeddings.position_embedding.weight"), B, S, d)


def _anchor_grad(*incoming):
    """A real (zero) gradient for the anchor leaf.  The towers write their parameter gradients straight into the flat buffers on
    the stream their forward ran on - autograd sees no AccumulateGrad for them, and it only joins the caller's stream with the
    streams of LEAF accumulations when ``backward()`` ends.  With the anchor's accumulation on this stream the join covers every
    kernel enqueued above: ``p.grad`` is safe to read (or to hand to ``torch.optim``) on the caller's stream as soon as
    ``loss.backward()`` returns, as in the reference.  (Returning None here left that to chance: a gradient read right after
    ``backward()`` raced with the last layers' kernels of the tower that ran alone on its stream.)"""
    ref = next((g for g in incoming if torch.is_tensor(g)), None)
    return torch.zeros((), device=ref.device if ref is not None else None)


class _TowerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, tower: ClipTower, inputs, need_grad, *params):
        state, last, pooled = tower._forward_impl(inputs, save=need_grad)
        ctx.tower, ctx.state, ctx.nparams = tower, state, len(params)
        # an output nobody differentiates through arrives as None in backward, not as a materialised zero tensor: the training loops read
        # the pooled output only, and a [N, S, d] fp32 zero gradient for last_hidden_state was 155 MB filled and copied per video step -
        # and hid from backward_lanes that the gradient enters the last layer on the CLS rows only
        ctx.set_materialize_grads(False)
        return last, pooled

    @staticmethod
    def backward(ctx, d_last, d_pooled):
        tower, state = ctx.tower, ctx.state
        if state is None or not state.save:
            raise RuntimeError("tower backward called twice or without saved activations")
        ctx.state = None
        tower._backward_impl(state, d_last, d_pooled)
        tower._grad_fresh = True
        if tower._post_backward is not None:
            tower._post_backward(tower)
        if ctx.nparams:
            st = tower._store
            grads = tuple(st.gview(n).clone() for n in tower._param_names if tower.is_trainable(n))
            return (_anchor_grad(d_last, d_pooled), None, None, None) + grads
        tower.attach_grads()
        return (_anchor_grad(d_last, d_pooled), None, None, None)


class _TowerGroupFn(torch.autograd.Function):
    """several shape-identical towers as ONE autograd node: forward and backward run in lock-step (forward_lanes / backward_lanes)"""

    @staticmethod
    def forward(ctx, anchor, towers, inputs, need_grad):
        outs = forward_lanes(towers, inputs, save=need_grad)
        ctx.towers, ctx.states = towers, [o[0] for o in outs]
        ctx.set_materialize_grads(False)     # (see _TowerFn.forward)
        flat = []
        for _, last, pooled in outs:
            flat += [last, pooled]
        return tuple(flat)

    @staticmethod
    def backward(ctx, *grads):
        towers, states = ctx.towers, ctx.states
        if states is None or not states[0].save:
            raise RuntimeError("tower-group backward called twice or without saved activations")
        ctx.states = None
        backward_lanes(towers, states, list(grads[0::2]), list(grads[1::2]))
        for t in towers:
            t._grad_fresh = True
            if t._post_backward is not None:
                t._post_backward(t)
            t.attach_grads()
        return (_anchor_grad(*grads), None, None, None)


def run_towers(towers, kwargs_list):
    """[(last_hidden_state, pooled_output)] of `towers[i](**kwargs_list[i])`; shape-identical vision towers run in lock-step with
    grouped GEMM launches (one tile grid for all of them), anything else one after the other - same results either way."""
    inputs = [(kw.get("pixel_values"), kw.get("patch_keep")) for kw in kwargs_list]
    grouped = (len(towers) > 1 and _GRAD_MODE == "direct" and all(i[0] is not None for i in inputs) and lanes_compatible(towers, inputs)
               and len({id(t) for t in towers}) == len(towers))
    if not grouped:
        return [t(**kw) for t, kw in zip(towers, kwargs_list)]
    for t in towers:
        t._ensure_ready()
    need_grad = torch.is_grad_enabled() and any(p.requires_grad for t in towers for p in t.parameters())
    flat = _TowerGroupFn.apply(towers[0]._anchor, list(towers), inputs, need_grad)
    return [(flat[2 * i], flat[2 * i + 1]) for i in range(len(towers))]
