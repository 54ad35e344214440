"""CPU oracle for the MissM-Benchmark forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``missm_benchmark_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker / reported baseline.

It is a plain-torch fp32 restatement (no ``transformers`` import, no ``nn.Module``)
of the reference algorithm.  Every function cites the reference lines it follows
(paths relative to /root/reference).  The arithmetic that the reference imports
from third-party ``transformers`` (``CLIPAttention``, ``CLIPMLP``,
``CLIPVisionEmbeddings``, ``CLIPTextEmbeddings``, ``_expand_mask``; call sites
languagebind/image/modeling_image.py:11-12,69,71,463,602) is unpinned upstream
(no requirements file); it is restated here in its transformers-4.3x form, which
is the API the reference calls (``causal_attention_mask=`` kwarg).

Pinning: ``oracle/make_golden.py`` runs the reference's own modules (imported
from /root/reference with the shims of ``oracle/ref_shims.py``) and stock
``transformers`` CLIP models built from local configs on seeded inputs, and
writes ``tests/golden/*.pt``; ``tests/test_oracle_golden.py`` checks this file
against those fixtures.  The reference itself has no tests / golden vectors
(SURVEY.md section 4), so these captured outputs are the pin.

Parameters are passed as flat ``dict[str, Tensor]`` keyed exactly like the
reference ``state_dict()`` of one tower (e.g. ``encoder.layers.0.self_attn.q_proj.weight``,
``pre_layrnorm.weight`` [sic]).  All functions are differentiable torch code, so
``torch.autograd`` on the oracle yields the reference gradients.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# ----------------------------------------------------------------------------
# configuration (shapes only; mirrors languagebind/image/configuration_image.py:70-105,181-232)
# ----------------------------------------------------------------------------
@dataclass
class VisionCfg:
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    num_channels: int = 3
    image_size: object = 224  # int or (height, width): resize_pos gives the audio model a (num_mel_bins, target_length) image
    patch_size: int = 16
    layer_norm_eps: float = 1e-5
    hidden_act: str = "quick_gelu"
    add_time_attn: bool = False
    num_frames: int = 1
    # image/audio/depth/thermal modeling files keep a temporal MLP in the time branch
    # (image/modeling_image.py:83-84,129-134); the video file removed it (video/modeling_video.py:189-190,235-240)
    temporal_mlp: bool = False
    force_patch_dropout: float = 0.0  # configuration_image.py:199; PatchDropout is active in training mode only
    # LoRA over the vision encoder (configuration_image.py:200-202; image/modeling_image.py:775-793); 0 = plain weights
    lora_r: int = 0
    lora_alpha: float = 16.0
    lora_dropout: float = 0.0

    @property
    def num_patches(self) -> int:
        hw = self.image_size
        h, w = (hw, hw) if isinstance(hw, int) else hw
        return (h // self.patch_size) * (w // self.patch_size)

    @property
    def seq_len(self) -> int:
        return self.num_patches + 1


@dataclass
class TextCfg:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    layer_norm_eps: float = 1e-5
    hidden_act: str = "quick_gelu"
    add_time_attn: bool = False  # configuration_image.py:105 hard-codes this
    num_frames: int = 1
    temporal_mlp: bool = False


# reference: src/model/baseline.py:8 ; depth/thermal are this build's documented extension
#  leaving codes 0-4 untouched.
MISSING_TYPE_INDEX = {"language": 1, "video": 2, "audio": 3, "image": 4, "depth": 5, "thermal": 6}


# ----------------------------------------------------------------------------
# third-party arithmetic restated (transformers 4.3x CLIP)
# ----------------------------------------------------------------------------
def activation(x: Tensor, name: str) -> Tensor:
    """``ACT2FN[config.hidden_act]``; default quick_gelu = x * sigmoid(1.702 x)
    (configuration_image.py:191; transformers/activations.py QuickGELUActivation)."""
    if name == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    if name == "gelu":
        return F.gelu(x)
    raise ValueError(f"unsupported hidden_act {name!r}")


def linear(x: Tensor, p: Params, name: str, lora_scaling: float = 0.0) -> Tensor:
    """``nn.Linear`` - or, where ``convert_to_lora`` (image/modeling_image.py:775-793) wrapped it, peft's ``lora.Linear`` forward
    restated from its published definition (peft is absent from this image and unpinned upstream: PARITY UNPINNED for the wrapper):
        result = base_layer(x) + lora_B(lora_A(lora_dropout(x))) * scaling,   scaling = lora_alpha / r
    with lora_dropout = 0.0 (the reference default).  Keys are the tower's state-dict keys with peft's wrapper prefixes removed
    (``normalize_peft_keys``): ``<name>.weight`` / ``.bias`` for the base layer, ``<name>.lora_A.default.weight`` [r, in] and
    ``<name>.lora_B.default.weight`` [out, r] for the adapter; a linear without adapter keys is a plain linear."""
    y = F.linear(x, p[name + ".weight"], p.get(name + ".bias"))
    a = p.get(name + ".lora_A.default.weight")
    if a is not None and lora_scaling:
        y = y + F.linear(F.linear(x, a), p[name + ".lora_B.default.weight"]) * lora_scaling
    return y


def normalize_peft_keys(sd: Params) -> Params:
    """peft state-dict keys (``encoder.base_model.model.layers.N.self_attn.q_proj.base_layer.weight``) -> the oracle's
    (``encoder.layers.N.self_attn.q_proj.weight``); adapter keys keep their ``lora_A.default.weight`` leaf"""
    return {k.replace(".base_model.model.", ".").replace(".base_layer.", "."): v for k, v in sd.items()}


def lora_scaling(cfg) -> float:
    r = getattr(cfg, "lora_r", 0)
    return float(getattr(cfg, "lora_alpha", 0.0)) / r if r else 0.0


def layer_norm(x: Tensor, p: Params, prefix: str, eps: float) -> Tensor:
    """``nn.LayerNorm(embed_dim, eps)`` sites: image/modeling_image.py:70,72,82,84,465,604,606."""
    return F.layer_norm(x, (x.shape[-1],), p[prefix + ".weight"], p[prefix + ".bias"], eps)


def clip_attention(x: Tensor, p: Params, prefix: str, num_heads: int,
                   attention_mask: Optional[Tensor] = None,
                   causal_attention_mask: Optional[Tensor] = None, lora: float = 0.0) -> Tensor:
    """``CLIPAttention.forward`` (third-party; called at image/modeling_image.py:121-126,140-145).

    q = (x Wq^T + bq) * hd^-1/2 ; scores = q k^T (+ causal) (+ padding) ; softmax over keys ;
    out = (P v) Wo^T + bo.  Masks are additive [B,1,S,S] with finfo.min at blocked positions.
    """
    bsz, tgt, dim = x.shape
    hd = dim // num_heads
    scale = hd ** -0.5
    q = linear(x, p, prefix + ".q_proj", lora) * scale
    k = linear(x, p, prefix + ".k_proj", lora)
    v = linear(x, p, prefix + ".v_proj", lora)

    def heads(t: Tensor) -> Tensor:
        return t.view(bsz, tgt, num_heads, hd).transpose(1, 2)  # [B,H,S,hd]

    q, k, v = heads(q), heads(k), heads(v)
    w = torch.matmul(q, k.transpose(-1, -2))  # [B,H,S,S]
    if causal_attention_mask is not None:
        w = w + causal_attention_mask
    if attention_mask is not None:
        w = w + attention_mask
    w = torch.softmax(w, dim=-1)
    o = torch.matmul(w, v)  # [B,H,S,hd]
    o = o.transpose(1, 2).reshape(bsz, tgt, dim)
    return linear(o, p, prefix + ".out_proj", lora)


def clip_mlp(x: Tensor, p: Params, prefix: str, act: str, lora: float = 0.0) -> Tensor:
    """``CLIPMLP.forward`` (third-party; called at image/modeling_image.py:150): fc2(act(fc1(x)))."""
    h = linear(x, p, prefix + ".fc1", lora)
    h = activation(h, act)
    return linear(h, p, prefix + ".fc2", lora)


def make_causal_mask(seq: int, dtype: torch.dtype) -> Tensor:
    """``_make_causal_mask`` image/modeling_image.py:441-455: finfo.min strictly above the diagonal."""
    m = torch.full((seq, seq), torch.finfo(dtype).min, dtype=dtype)
    return torch.triu(m, diagonal=1)[None, None]


def expand_padding_mask(attention_mask: Tensor, dtype: torch.dtype) -> Tensor:
    """``_expand_mask`` (third-party, used at image/modeling_image.py:499-501):
    [B,S] 1/0 -> additive [B,1,S,S] with finfo.min on masked key columns."""
    bsz, src = attention_mask.shape
    expanded = attention_mask[:, None, None, :].expand(bsz, 1, src, src).to(dtype)
    inverted = 1.0 - expanded
    return inverted.masked_fill(inverted.to(torch.bool), torch.finfo(dtype).min)


# ----------------------------------------------------------------------------
# encoder layer / encoder  (image/modeling_image.py:86-158,352-437; video/modeling_video.py:192-264)
# ----------------------------------------------------------------------------
def encoder_layer(h: Tensor, p: Params, prefix: str, cfg, attention_mask=None, causal_attention_mask=None) -> Tensor:
    """``CLIPEncoderLayer.forward``.  ``h`` is [(b t), n, d] for vision, [b, s, d] for text."""
    nh = cfg.num_attention_heads
    eps = cfg.layer_norm_eps
    ls = lora_scaling(cfg)      # adapters exist only on the linears convert_to_lora targets (:778-783); `linear` looks for their keys
    if cfg.add_time_attn:
        bt, n, d = h.shape
        t = cfg.num_frames
        b = bt // t
        if t != 1:  # time embed (:110-114)
            hh = h.view(b, t, n, d).permute(0, 2, 1, 3).reshape(b * n, t, d)
            hh = hh + p[prefix + ".temporal_embedding"][:, :t, :]
            h = hh.view(b, n, t, d).permute(0, 2, 1, 3).reshape(bt, n, d)
        # time attn (:117-127)
        residual = h
        hh = h.view(b, t, n, d).permute(0, 2, 1, 3).reshape(b * n, t, d)
        hh = layer_norm(hh, p, prefix + ".temporal_layer_norm1", eps)
        hh = clip_attention(hh, p, prefix + ".temporal_attn", nh, attention_mask, causal_attention_mask, lora=ls)
        h = residual + hh.view(b, n, t, d).permute(0, 2, 1, 3).reshape(bt, n, d)
        if cfg.temporal_mlp:  # image-family only (:129-134)
            residual = h
            hh = h.view(b, t, n, d).permute(0, 2, 1, 3).reshape(b * n, t, d)
            hh = layer_norm(hh, p, prefix + ".temporal_layer_norm2", eps)
            hh = clip_mlp(hh, p, prefix + ".temporal_mlp", cfg.hidden_act, lora=ls)
            h = residual + hh.view(b, n, t, d).permute(0, 2, 1, 3).reshape(bt, n, d)
    # spatial attn (:137-146)
    residual = h
    x = layer_norm(h, p, prefix + ".layer_norm1", eps)
    x = clip_attention(x, p, prefix + ".self_attn", nh, attention_mask, causal_attention_mask, lora=ls)
    h = residual + x
    # mlp (:148-151)
    residual = h
    x = layer_norm(h, p, prefix + ".layer_norm2", eps)
    x = clip_mlp(x, p, prefix + ".mlp", cfg.hidden_act, lora=ls)
    return residual + x


def encoder(h: Tensor, p: Params, cfg, attention_mask=None, causal_attention_mask=None) -> Tensor:
    """``CLIPEncoder.forward`` image/modeling_image.py:400-428 (no grad-checkpointing, hooks off)."""
    for i in range(cfg.num_hidden_layers):
        h = encoder_layer(h, p, f"encoder.layers.{i}", cfg, attention_mask, causal_attention_mask)
    return h


# ----------------------------------------------------------------------------
# towers
# ----------------------------------------------------------------------------
def vision_embeddings(pixel_values: Tensor, p: Params, cfg: VisionCfg) -> Tensor:
    """``CLIPVisionEmbeddings.forward`` video/modeling_video.py:42-51 (third-party twin for the other towers):
    bias-free conv k=stride=patch -> flatten -> cat CLS -> + position embedding."""
    n = pixel_values.shape[0]
    pe = F.conv2d(pixel_values, p["embeddings.patch_embedding.weight"], None, stride=cfg.patch_size)
    pe = pe.flatten(2).transpose(1, 2)  # [N, P, d]
    cls = p["embeddings.class_embedding"].expand(n, 1, -1)
    x = torch.cat([cls, pe], dim=1)
    return x + p["embeddings.position_embedding.weight"][None, : x.shape[1]]


def patch_dropout(x: Tensor, keep: Tensor, B: int, T: int) -> Tensor:
    """``PatchDropout.forward`` image/modeling_image.py:30-63 in training mode with prob > 0, the random draw handed in:
    ``keep`` = ``torch.randn(batch, num_tokens).topk(num_patches_keep).indices`` ([(B T), K] for T == 1, else [B, K] shared by a
    sample's frames); the class token is excluded from the draw and kept."""
    cls_tokens, x = x[:, :1], x[:, 1:]
    if T != 1:
        keep = keep.unsqueeze(1).repeat(1, T, 1).reshape(B * T, -1)         # 'b t n -> (b t) n'
    x = x[torch.arange(x.shape[0])[:, None], keep]
    return torch.cat((cls_tokens, x), dim=1)


def vision_tower(pixel_values: Tensor, p: Params, cfg: VisionCfg, patch_keep: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """``CLIPVisionTransformer.forward`` image/modeling_image.py:610-672 / video/modeling_video.py:723-784.
    Returns (last_hidden_state [(B T), S, d], pooled [B, d]).  ``patch_keep``: the kept-token indices of a training-mode
    PatchDropout (None: eval mode / prob 0 = identity)."""
    if pixel_values is None:
        raise ValueError("You have to specify pixel_values")
    if pixel_values.dim() == 7:  # :630-634
        b_new, pair_new, T, bs_new, c, hh, ww = pixel_values.shape
        B = b_new * pair_new * bs_new
        pixel_values = pixel_values.reshape(B * T, c, hh, ww)
    elif pixel_values.dim() == 5:  # :636-639  'b c t h w -> (b t) c h w'
        B, c, T, hh, ww = pixel_values.shape
        pixel_values = pixel_values.permute(0, 2, 1, 3, 4).reshape(B * T, c, hh, ww)
    else:
        B, T = pixel_values.shape[0], 1
    h = vision_embeddings(pixel_values, p, cfg)
    # PatchDropout (:30-63,647) is the identity at force_patch_dropout == 0 (default) and in eval.
    if patch_keep is not None:
        h = patch_dropout(h, patch_keep, B, T)
    h = layer_norm(h, p, "pre_layrnorm", cfg.layer_norm_eps)
    h = encoder(h, p, cfg)
    pooled = layer_norm(h[:, 0, :], p, "post_layernorm", cfg.layer_norm_eps)
    pooled = pooled.reshape(B, T, -1).mean(1)  # :662
    return h, pooled


def text_tower(input_ids: Tensor, attention_mask: Optional[Tensor], p: Params, cfg: TextCfg) -> Tuple[Tensor, Tensor]:
    """``CLIPTextTransformer.forward`` image/modeling_image.py:469-532.
    Returns (last_hidden_state [B,S,d] after final LN, pooled [B,d] at the EOT position)."""
    if input_ids is None:
        raise ValueError("You have to specify input_ids")
    input_ids = input_ids.view(-1, input_ids.shape[-1])
    bsz, seq = input_ids.shape
    tok = p["embeddings.token_embedding.weight"][input_ids]          # CLIPTextEmbeddings (third-party)
    h = tok + p["embeddings.position_embedding.weight"][None, :seq]
    causal = make_causal_mask(seq, h.dtype)
    pad = expand_padding_mask(attention_mask, h.dtype) if attention_mask is not None else None
    h = encoder(h, p, cfg, attention_mask=pad, causal_attention_mask=causal)
    h = layer_norm(h, p, "final_layer_norm", cfg.layer_norm_eps)
    eot = input_ids.to(torch.int).argmax(dim=-1)                      # :519-522
    pooled = h[torch.arange(bsz), eot]
    return h, pooled


# ----------------------------------------------------------------------------
# bundle + fusion + loss
# ----------------------------------------------------------------------------
def bundle_embed(pooled: Tensor, proj_weight: Tensor, logit_scale: Optional[Tensor], modality: str,
                 use_temp: bool = True) -> Tensor:
    """``LanguageBind.forward`` body, languagebind/__init__.py:78-84:
    bias-free projection -> / L2 norm -> * exp(logit_scale) unless language."""
    v = F.linear(pooled, proj_weight)
    v = v / v.norm(p=2, dim=-1, keepdim=True)
    if use_temp and modality != "language":
        v = v * logit_scale.exp()
    return v


def head_forward(x: Tensor, fp: Params, prefix: str = "head.head") -> Tensor:
    """``Head`` src/model/baseline.py:27-39 with Dropout in eval / p=0 (identity)."""
    x = F.linear(x, fp[prefix + ".0.weight"], fp[prefix + ".0.bias"])
    x = F.relu(x)
    return F.linear(x, fp[prefix + ".3.weight"], fp[prefix + ".3.bias"])


def fusion_sum(emb: Dict[str, Tensor], missing_index: Tensor, fp: Params, modality_types: Sequence[str],
               codes: Dict[str, int] = MISSING_TYPE_INDEX) -> Tensor:
    """``modal_sum.forward`` src/model/baseline.py:52-61: per-modality Linear, zero rows whose
    missing code matches, sum, LayerNorm(eps 1e-5), Head."""
    z = None
    for m in modality_types:
        d = F.linear(emb[m], fp[f"modal_proj.{m}.weight"], fp[f"modal_proj.{m}.bias"])
        d = torch.where((missing_index == codes[m])[:, None], torch.zeros_like(d), d)
        z = d if z is None else z + d
    z = F.layer_norm(z, (z.shape[-1],), fp["norm.weight"], fp["norm.bias"], 1e-5)
    return head_forward(z, fp)


def fusion_concat(emb: Dict[str, Tensor], missing_index: Tensor, fp: Params, modality_types: Sequence[str],
                  statistics: Optional[Dict[str, Tensor]] = None, mask: bool = True,
                  codes: Dict[str, int] = MISSING_TYPE_INDEX) -> Tensor:
    """``modal_concat.forward`` src/model/baseline.py:77-86 (mask=True: rows of a missing modality are overwritten with its
    ``statistics_<modal>`` buffer - zeros unless ``set_statistics`` ran, :74-75,88-90 - BEFORE the projection) and
    ``modal_concat_full.forward`` :172-180 (mask=False); projections concatenated, LayerNorm, Head."""
    parts = []
    for m in modality_types:
        x = emb[m]
        if mask:
            st = statistics[m] if statistics is not None else torch.zeros(x.shape[-1])
            x = torch.where((missing_index == codes[m])[:, None], st[None].to(x.dtype), x)
        parts.append(F.linear(x, fp[f"modal_proj.{m}.weight"], fp[f"modal_proj.{m}.bias"]))
    z = torch.cat(parts, dim=-1)
    z = F.layer_norm(z, (z.shape[-1],), fp["norm.weight"], fp["norm.bias"], 1e-5)
    return head_forward(z, fp)


def fusion_intra_attention(emb: Dict[str, Tensor], missing_index: Tensor, fp: Params, modality_types: Sequence[str],
                           codes: Dict[str, int] = MISSING_TYPE_INDEX) -> Tensor:
    """``modal_intra_channel_attention.forward`` src/model/baseline.py:183-205: per modality d = Linear(x);
    gate = sigmoid(Linear(relu(Linear([d | fusion_representation])))); d * gate, missing rows zeroed; sum; LayerNorm; Head."""
    z = None
    rep = fp["fusion_representation"]
    for m in modality_types:
        d = F.linear(emb[m], fp[f"modal_proj.{m}.weight"], fp[f"modal_proj.{m}.bias"])
        c = torch.cat([d, rep.expand(d.shape[0], -1)], dim=-1)
        h = F.relu(F.linear(c, fp["channel_attention.0.weight"], fp["channel_attention.0.bias"]))
        g = torch.sigmoid(F.linear(h, fp["channel_attention.2.weight"], fp["channel_attention.2.bias"]))
        y = d * g
        y = torch.where((missing_index == codes[m])[:, None], torch.zeros_like(y), y)
        z = y if z is None else z + y
    z = F.layer_norm(z, (z.shape[-1],), fp["norm.weight"], fp["norm.bias"], 1e-5)
    return head_forward(z, fp)


def fusion_inter_attention(emb: Dict[str, Tensor], missing_index: Tensor, fp: Params, modality_types: Sequence[str],
                           codes: Dict[str, int] = MISSING_TYPE_INDEX, num_heads: int = 4) -> Tensor:
    """``modal_inter_attention.forward`` src/model/baseline.py:207-236: the projected modalities are M tokens; a learned query
    token attends over them with ``nn.MultiheadAttention(fusion_dim, 4, batch_first=True)`` (third-party torch arithmetic,
    restated: packed in-projection, q scaled by head_dim^-1/2, missing modalities removed through ``key_padding_mask`` =
    -inf scores, softmax over the M keys, out-projection); LayerNorm; Head."""
    tokens = torch.stack([F.linear(emb[m], fp[f"modal_proj.{m}.weight"], fp[f"modal_proj.{m}.bias"]) for m in modality_types], dim=1)
    pad = torch.stack([missing_index == codes[m] for m in modality_types], dim=1)               # [B, M], True = ignore
    B, M, D = tokens.shape
    hd = D // num_heads
    w, b = fp["attn.in_proj_weight"], fp["attn.in_proj_bias"]
    q = F.linear(fp["query_token"].expand(B, -1, -1), w[:D], b[:D])                              # [B, 1, D]
    k = F.linear(tokens, w[D:2 * D], b[D:2 * D])
    v = F.linear(tokens, w[2 * D:], b[2 * D:])
    qh = q.view(B, 1, num_heads, hd).transpose(1, 2) * hd ** -0.5
    kh = k.view(B, M, num_heads, hd).transpose(1, 2)
    vh = v.view(B, M, num_heads, hd).transpose(1, 2)
    scores = qh @ kh.transpose(-1, -2)                                                           # [B, H, 1, M]
    scores = scores.masked_fill(pad[:, None, None, :], float("-inf"))
    ctx = (torch.softmax(scores, dim=-1) @ vh).transpose(1, 2).reshape(B, 1, D)
    out = F.linear(ctx, fp["attn.out_proj.weight"], fp["attn.out_proj.bias"])[:, 0, :]
    z = F.layer_norm(out, (D,), fp["norm.weight"], fp["norm.bias"], 1e-5)
    return head_forward(z, fp)


def supergat_conv(x: Tensor, node_ok: Tensor, fp: Params, prefix: str, heads: int, concat: bool, negative_slope: float = 0.2) -> Tensor:
    """``torch_geometric.nn.SuperGATConv(in, out, heads, concat)`` (attention_type 'MX', add_self_loops) as ``fusion_gcn`` uses it
    (src/model/baseline.py:14-15,19-21), on the dense per-sample modality graphs the heads build (``bulid_edge`` :270-277: edges in
    both directions between modalities that are present; the convolution adds self loops).  PARITY UNPINNED: torch_geometric is
    absent from this image and unpinned upstream; restated from the published algorithm (Kim & Oh, ICLR 2021, 'MX' attention:
    e_ij = (a_l . Wx_j + a_r . Wx_i) * sigmoid(Wx_i . Wx_j), leaky_relu, softmax over the incoming edges, sum_j alpha_ij Wx_j, bias).
    x [B, M, in], node_ok bool [B, M] -> [B, M, heads * out] (concat) or [B, M, out] (mean over heads)."""
    B, M, _ = x.shape
    W, al, ar, bias = fp[prefix + ".lin.weight"], fp[prefix + ".att_l"], fp[prefix + ".att_r"], fp[prefix + ".bias"]
    C = W.shape[0] // heads
    xp = F.linear(x, W).view(B, M, heads, C)
    logits = torch.einsum("bihc,bjhc->bijh", xp, xp)
    a = (xp * al).sum(-1)[:, None, :, :] + (xp * ar).sum(-1)[:, :, None, :]          # [B, i (target), j (source), H]
    a = F.leaky_relu(a * torch.sigmoid(logits), negative_slope)
    eye = torch.eye(M, dtype=torch.bool)[None]
    edge = eye | (node_ok[:, :, None] & node_ok[:, None, :])
    a = a.masked_fill(~edge[..., None], float("-inf"))
    alpha = torch.softmax(a, dim=2)
    out = torch.einsum("bijh,bjhc->bihc", alpha, xp)
    out = out.reshape(B, M, heads * C) if concat else out.mean(dim=2)
    return out + bias


def fusion_gcn(x: Tensor, node_ok: Tensor, fp: Params, prefix: str, heads: int = 4) -> Tensor:
    """``fusion_gcn.forward`` src/model/baseline.py:18-24: SuperGAT (4 heads, concat) -> exact GELU -> SuperGAT (1 head, mean)."""
    h = F.gelu(supergat_conv(x, node_ok, fp, prefix + ".gat1", heads, True))
    return supergat_conv(h, node_ok, fp, prefix + ".gat2", 1, False)


def fusion_graph(emb: Dict[str, Tensor], missing_index: Tensor, fp: Params, modality_types: Sequence[str],
                 codes: Dict[str, int] = MISSING_TYPE_INDEX) -> Tensor:
    """``modal_graph_fusion.forward`` src/model/baseline.py:253-268 (PARITY UNPINNED through supergat_conv)."""
    x = torch.stack([F.linear(emb[m], fp[f"modal_proj.{m}.weight"], fp[f"modal_proj.{m}.bias"]) for m in modality_types], dim=1)
    ok = torch.stack([missing_index != codes[m] for m in modality_types], dim=1)
    z = fusion_gcn(x, ok, fp, "gcn").mean(dim=-2)
    z = F.layer_norm(z, (z.shape[-1],), fp["norm.weight"], fp["norm.bias"], 1e-5)
    return head_forward(z, fp)


def fusion_unified_graph(emb: Dict[str, Tensor], missing_index: Tensor, fp: Params, modality_types: Sequence[str],
                         codes: Dict[str, int] = MISSING_TYPE_INDEX) -> Tensor:
    """``modal_unified_graph.forward`` src/model/baseline.py:291-324 (PARITY UNPINNED through supergat_conv): a completion network
    fills the missing modality's embedding, the fusion network runs on the complete graph."""
    x = torch.stack([emb[m] for m in modality_types], dim=1)
    miss = torch.stack([missing_index == codes[m] for m in modality_types], dim=1)
    done = fusion_gcn(x, ~miss, fp, "complete_gcn")
    x = torch.where(miss[..., None], done, x)
    z = fusion_gcn(x, torch.ones_like(miss), fp, "fusion_gcn").mean(dim=-2)
    z = F.layer_norm(z, (z.shape[-1],), fp["norm.weight"], fp["norm.bias"], 1e-5)
    return head_forward(z, fp)


def fusion_dedicated_dnn(emb: Dict[str, Tensor], missing_index: Tensor, fp: Params, modality_types: Sequence[str],
                         codes: Dict[str, int] = MISSING_TYPE_INDEX) -> Tensor:
    """``modal_dedicated_dnn.forward`` src/model/baseline.py:345-353: full network on the concatenated embeddings; rows whose
    modality i is missing are overwritten by dedicated_i over the concatenation without block i; LayerNorm; Head."""
    feats = [emb[m] for m in modality_types]
    z = F.linear(torch.cat(feats, dim=-1), fp["dedicated_dnn.full.weight"], fp["dedicated_dnn.full.bias"])
    for i, m in enumerate(modality_types):
        wo = torch.cat(feats[:i] + feats[i + 1:], dim=-1)
        zi = F.linear(wo, fp[f"dedicated_dnn.{m}.weight"], fp[f"dedicated_dnn.{m}.bias"])
        z = torch.where((missing_index == codes[m])[:, None], zi, z)
    z = F.layer_norm(z, (z.shape[-1],), fp["norm.weight"], fp["norm.bias"], 1e-5)
    return head_forward(z, fp)


def fusion_regression(emb: Dict[str, Tensor], missing_index: Tensor, fp: Params, modality_types: Sequence[str],
                      codes: Dict[str, int] = MISSING_TYPE_INDEX) -> Tensor:
    """``modal_regression.forward`` src/model/baseline.py:113-161: projections; where target t is missing, the mean over the
    sources s != t that are present of ``cross_modal_regressors[s_to_t](x_s)`` replaces the projection; concat; LayerNorm; Head."""
    proj = {m: F.linear(emb[m], fp[f"modal_proj.{m}.weight"], fp[f"modal_proj.{m}.bias"]) for m in modality_types}
    for t in modality_types:
        tmask = missing_index == codes[t]
        preds, masks = [], []
        for s in modality_types:
            if s == t:
                continue
            preds.append(F.linear(emb[s], fp[f"cross_modal_regressors.{s}_to_{t}.weight"], fp[f"cross_modal_regressors.{s}_to_{t}.bias"]))
            masks.append((missing_index != codes[s]).to(preds[-1].dtype))
        p = torch.stack(preds, dim=1) * torch.stack(masks, dim=-1).unsqueeze(-1)
        avg = p.sum(dim=1) / torch.stack(masks, dim=-1).unsqueeze(-1).sum(dim=1).clamp(min=1e-6)
        proj[t] = torch.where(tmask[:, None], avg, proj[t])
    z = torch.cat([proj[m] for m in modality_types], dim=-1)
    z = F.layer_norm(z, (z.shape[-1],), fp["norm.weight"], fp["norm.bias"], 1e-5)
    return head_forward(z, fp)


def fusion_distillation(emb: Dict[str, Tensor], missing_index: Tensor, fp: Params, modality_types: Sequence[str],
                        codes: Dict[str, int] = MISSING_TYPE_INDEX) -> Tuple[Tensor, Tensor]:
    """``modal_distillation.forward`` src/model/baseline.py:370-380: zero the missing modality, concatenate, Linear-ReLU-Linear,
    LayerNorm, Head; returns (features, logits)."""
    feats = torch.cat([torch.where((missing_index == codes[m])[:, None], torch.zeros_like(emb[m]), emb[m]) for m in modality_types], dim=-1)
    h = F.relu(F.linear(feats, fp["modal_proj.0.weight"], fp["modal_proj.0.bias"]))
    h = F.linear(h, fp["modal_proj.2.weight"], fp["modal_proj.2.bias"])
    z = F.layer_norm(h, (h.shape[-1],), fp["norm.weight"], fp["norm.bias"], 1e-5)
    return feats, head_forward(z, fp)


def fusion_self_distillation(emb: Dict[str, Tensor], missing_index: Tensor, fp: Params, modality_types: Sequence[str],
                             codes: Dict[str, int] = MISSING_TYPE_INDEX):
    """``modal_self_distillation.forward`` in training mode, src/model/baseline.py:397-411:
    (missing_mask list, student features per modality, teacher features, logits)."""
    def proj(f):
        return F.linear(F.relu(F.linear(f, fp["modal_proj.0.weight"], fp["modal_proj.0.bias"])), fp["modal_proj.2.weight"], fp["modal_proj.2.bias"])
    masked = [torch.where((missing_index == codes[m])[:, None], torch.zeros_like(emb[m]), emb[m]) for m in modality_types]
    stu, masks = [], []
    for i, m in enumerate(modality_types):
        stu.append(proj(torch.cat([masked[j] if j == i else torch.zeros_like(masked[j]) for j in range(len(masked))], dim=-1)))
        masks.append(missing_index != codes[m])
    tea = proj(torch.cat(masked, dim=-1))
    z = F.layer_norm(tea, (tea.shape[-1],), fp["norm.weight"], fp["norm.bias"], 1e-5)
    return masks, stu, tea, head_forward(z, fp)


def cross_entropy(logits: Tensor, labels: Tensor) -> Tensor:
    """``nn.CrossEntropyLoss()`` train_ddp.py:88,250 (mean over batch)."""
    return F.cross_entropy(logits, labels)


def resize_pos_embed(weight: Tensor, grid: Sequence[int], extra_tokens: int = 1) -> Tensor:
    """``resize_pos`` image/modeling_image.py:795-839: class-token rows kept, the square patch grid resampled to ``grid`` with
    the reference's own third-party call ``F.interpolate(mode='bicubic', antialias=True, align_corners=False)``."""
    if grid[0] * grid[1] + extra_tokens == weight.shape[0]:
        return weight
    tok, img = weight[:extra_tokens], weight[extra_tokens:]
    old = int(math.sqrt(len(img)))
    img = img.reshape(1, old, old, -1).permute(0, 3, 1, 2)
    img = F.interpolate(img, size=list(grid), mode="bicubic", antialias=True, align_corners=False)
    img = img.permute(0, 2, 3, 1).reshape(1, grid[0] * grid[1], -1)[0]
    return torch.cat([tok, img], dim=0)


OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)


def image_transform(img_chw: Tensor, size: int = 224) -> Tensor:
    """``get_image_transform`` image/processing_image.py:18-28 after ToTensor (img_chw float [C,H,W] in [0,1]):
    torchvision ``Resize(size, BICUBIC)`` on a tensor = ``F.interpolate(mode='bicubic', antialias=True, align_corners=False)`` of the
    shorter edge to ``size`` (longer edge int(size * long / short)); ``CenterCrop`` (offsets int(round((n - size) / 2)));
    ``Normalize``.  torchvision is absent from this image and unpinned upstream: parity unpinned (restated from its source)."""
    c, h, w = img_chw.shape
    nh, nw = (size, int(size * w / h)) if h <= w else (int(size * h / w), size)
    r = F.interpolate(img_chw[None], size=[nh, nw], mode="bicubic", antialias=True, align_corners=False)[0]
    top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))
    r = r[:, top:top + size, left:left + size]
    mean, std = torch.tensor(OPENAI_DATASET_MEAN)[:, None, None], torch.tensor(OPENAI_DATASET_STD)[:, None, None]
    return (r - mean) / std


def depth_transform(depth_hw: Tensor, max_depth: float = 10.0, size: int = 224) -> Tensor:
    """``get_depth_transform`` depth/processing_depth.py:21-55: DepthNorm (/1000, clip to [0.01, max_depth], / max_depth, repeated to
    3 channels) followed by the image pipeline."""
    d = (depth_hw / 1000.0).clip(min=0.01)
    d = d.clip(max=max_depth) / max_depth if max_depth != 0 else d / d.max()
    return image_transform(d[None].repeat(3, 1, 1), size)


# ----------------------------------------------------------------------------
# audio front end (languagebind/audio/processing_audio.py:31-111).  torchaudio is ABSENT from this image and unpinned upstream:
# ``torchaudio.compliance.kaldi.fbank`` and ``torchaudio.functional.resample`` are restated here from their published algorithms in
# plain torch - PARITY UNPINNED for these two (no reference run, no fixtures upstream); the reference's own code around them
# (mean removal, chunking, normalisation) is followed line by line.
# ----------------------------------------------------------------------------
def kaldi_fbank(waveform: Tensor, sample_frequency: float, num_mel_bins: int, frame_length: float = 25.0, frame_shift: float = 10.0,
                low_freq: float = 20.0, high_freq: float = 0.0, preemphasis_coefficient: float = 0.97) -> Tensor:
    """``torchaudio.compliance.kaldi.fbank(waveform, htk_compat=True, sample_frequency=..., use_energy=False, window_type="hanning",
    num_mel_bins=..., dither=0.0, frame_length=25, frame_shift=10)`` as called at processing_audio.py:97-107, remaining arguments at
    their defaults (channel 0, snip_edges, remove_dc_offset, preemphasis 0.97, round_to_power_of_two, use_power, use_log_fbank,
    energy_floor unused, no vtln warp): frames of ``waveform[0]`` -> DC removal -> pre-emphasis (first sample against itself) -> hann
    window (non-periodic) -> zero pad to a power of two -> |rfft|^2 -> triangular mel filters (mel = 1127 ln(1 + f / 700), zero
    weight on the Nyquist bin) -> log(max(., eps)).  Returns [frames, num_mel_bins]."""
    wave = waveform[0] if waveform.dim() == 2 else waveform
    window_shift = int(sample_frequency * 0.001 * frame_shift)
    window_size = int(sample_frequency * 0.001 * frame_length)
    padded = 1 << (window_size - 1).bit_length()
    if wave.numel() < window_size:
        return torch.empty(0, num_mel_bins)
    m = 1 + (wave.numel() - window_size) // window_shift
    frames = wave.as_strided((m, window_size), (window_shift, 1))
    frames = frames - frames.mean(dim=1, keepdim=True)
    prev = F.pad(frames.unsqueeze(0), (1, 0), mode="replicate").squeeze(0)[:, :-1]
    frames = frames - preemphasis_coefficient * prev
    frames = frames * torch.hann_window(window_size, periodic=False, dtype=frames.dtype).unsqueeze(0)
    frames = F.pad(frames, (0, padded - window_size))
    spectrum = torch.fft.rfft(frames).abs().pow(2.0)                               # [m, padded / 2 + 1]
    nyquist = 0.5 * sample_frequency
    if high_freq <= 0.0:
        high_freq += nyquist
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)                             # noqa: E731
    mel_low, mel_high = mel(low_freq), mel(high_freq)
    delta = (mel_high - mel_low) / (num_mel_bins + 1)
    b = torch.arange(num_mel_bins, dtype=torch.float32).unsqueeze(1)
    left, center, right = mel_low + b * delta, mel_low + (b + 1.0) * delta, mel_low + (b + 2.0) * delta
    fft_mel = (1127.0 * (1.0 + (sample_frequency / padded) * torch.arange(padded // 2, dtype=torch.float32) / 700.0).log()).unsqueeze(0)
    bank = torch.max(torch.zeros(1), torch.min((fft_mel - left) / (center - left), (right - fft_mel) / (right - center)))
    bank = F.pad(bank, (0, 1))
    energies = spectrum @ bank.T
    return torch.max(energies, torch.tensor(torch.finfo(torch.float32).eps)).log()


def sinc_resample(waveform: Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> Tensor:
    """``torchaudio.functional.resample(waveform, orig_freq, new_freq)`` (processing_audio.py:46) with its defaults (``sinc_interp_hann``,
    lowpass_filter_width 6, rolloff 0.99): a strided convolution with ``new`` windowed-sinc kernels (frequencies divided by their gcd)."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = (torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx) * base
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * (base / orig)
    kernels = kernels.to(torch.float32)
    shape = waveform.shape
    wave = waveform.reshape(-1, shape[-1])
    length = wave.shape[1]
    wave = F.pad(wave, (width, width + orig))
    out = F.conv1d(wave[:, None], kernels, stride=orig).transpose(1, 2).reshape(wave.shape[0], -1)
    target = int(math.ceil(new * length / orig))
    return out[..., :target].reshape(shape[:-1] + (target,))


def audio_transform(audio_data: Tensor, origin_sr: int, sample_rate: int, num_mel_bins: int, target_length: int, audio_mean: float,
                    audio_std: float, starts: Optional[Sequence[int]] = None) -> Tensor:
    """``AudioTransform.__call__`` / ``waveform2melspec`` / ``get_mel`` processing_audio.py:41-108.  ``starts``: the three chunk starts
    (the reference draws them with ``np.random.choice``, :69-71); needed only when the clip has more than target_length frames."""
    if sample_rate != origin_sr:
        audio_data = sinc_resample(audio_data, origin_sr, sample_rate)
    audio_data = audio_data - audio_data.mean()                                    # :96
    mel = kaldi_fbank(audio_data, sample_rate, num_mel_bins)                       # (T, n_mels)
    if mel.shape[0] > target_length:                                               # :56-77
        mel_fusion = torch.stack([mel[s:s + target_length] for s in starts], dim=0)
    elif mel.shape[0] < target_length:                                             # :78-82
        n_repeat = int(target_length / mel.shape[0]) + 1
        mel = mel.repeat(n_repeat, 1)[:target_length]
        mel_fusion = torch.stack([mel, mel, mel], dim=0)
    else:
        mel_fusion = torch.stack([mel, mel, mel], dim=0)
    mel_fusion = mel_fusion.transpose(1, 2)                                        # [3, mel_bins, target_length]
    return (mel_fusion - audio_mean) / (audio_std * 2)                             # :92


def kl_loss(g_s: Tensor, g_t: Tensor, temperature: float = 0.15) -> Tensor:
    """``KL_loss.forward`` train_ddp.py:70-79: kl_div(log_softmax(g_s / T), softmax(g_t.detach() / T), reduction='batchmean')."""
    return F.kl_div(F.log_softmax(g_s / temperature, dim=1), F.softmax(g_t.detach() / temperature, dim=1), reduction="batchmean")


def mse_loss(a: Tensor, b: Tensor) -> Tensor:
    """``nn.MSELoss()`` of the MTD student mode (train_ddp.py:84)."""
    return F.mse_loss(a, b)


def self_distill_loss(masks: Sequence[Tensor], stu: Sequence[Tensor], tea: Tensor, logits: Tensor, labels: Tensor,
                      temperature: float = 0.15) -> Tensor:
    """the self-distillation branch of the training loop, train_ddp.py:235-242:
    0.01 * mean_i KL(stu_i[mask_i], tea[mask_i]) + CE(logits, labels)."""
    dl = 0
    for i, mask in enumerate(masks):
        dl = dl + kl_loss(stu[i][mask], tea[mask], temperature)
    return 0.01 * dl / len(masks) + cross_entropy(logits, labels)


def ema_update(tea: Tensor, stu: Tensor, decay: float = 0.999) -> Tensor:
    """teacher EMA of the MTD student mode, train_ddp.py:256-259."""
    return tea * decay + stu * (1.0 - decay)


def finetune_forward(data: Dict[str, Dict[str, Tensor]], missing_index: Tensor, tower_params: Dict[str, Params],
                     tower_cfgs: Dict[str, object], proj: Dict[str, Tensor], scales: Dict[str, Tensor],
                     fusion_params: Params, modality_types: Sequence[str]) -> Tuple[Tensor, Dict[str, Tensor]]:
    """``finetune_model.forward`` src/model/baseline.py:450-453 with fusion_type == 'sum'.
    Returns (logits, bundle embeddings)."""
    emb: Dict[str, Tensor] = {}
    for m, inputs in data.items():
        if m == "language":
            _, pooled = text_tower(inputs["input_ids"], inputs.get("attention_mask"), tower_params[m], tower_cfgs[m])
        else:
            _, pooled = vision_tower(inputs["pixel_values"], tower_params[m], tower_cfgs[m])
        emb[m] = bundle_embed(pooled, proj[m], scales.get(m), m)
    return fusion_sum(emb, missing_index, fusion_params, modality_types), emb


# ----------------------------------------------------------------------------
# Adam (torch.optim.Adam defaults, train_ddp.py:205): one step on a flat tensor
# ----------------------------------------------------------------------------
def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float = 0.9,
              beta2: float = 0.999, eps: float = 1e-8, weight_decay: float = 0.0) -> None:
    """In-place single Adam update identical to ``torch.optim.Adam`` (non-amsgrad, L2 weight decay)."""
    if weight_decay != 0.0:
        g = g + weight_decay * p
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# ----------------------------------------------------------------------------
# seeded synthetic parameters / inputs  (the build's own recipe)
# ----------------------------------------------------------------------------
def _normal(shape, std: float, gen: torch.Generator) -> Tensor:
    return torch.randn(shape, generator=gen, dtype=torch.float32) * std


def init_tower_params(cfg, seed: int, kind: str = "vision") -> Params:
    """Deterministic per-tensor init with the std's of ``CLIPPreTrainedModel._init_weights``
    (image/modeling_image.py:179-230) but an RNG order owned by this build.  Biases and LN
    shifts get small non-zero values so that bias/shift paths are exercised by parity tests."""
    gen = torch.Generator().manual_seed(seed)
    d, f, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    p: Params = {}
    in_std = d ** -0.5 * (2 * L) ** -0.5
    out_std = d ** -0.5
    fc_std = (2 * d) ** -0.5

    def ln(name):
        p[name + ".weight"] = 1.0 + _normal((d,), 0.02, gen)
        p[name + ".bias"] = _normal((d,), 0.02, gen)

    def attn(name):
        for w in ("q_proj", "k_proj", "v_proj"):
            p[f"{name}.{w}.weight"] = _normal((d, d), in_std, gen)
            p[f"{name}.{w}.bias"] = _normal((d,), 0.02, gen)
        p[f"{name}.out_proj.weight"] = _normal((d, d), out_std, gen)
        p[f"{name}.out_proj.bias"] = _normal((d,), 0.02, gen)

    def mlp(name):
        p[f"{name}.fc1.weight"] = _normal((f, d), fc_std, gen)
        p[f"{name}.fc1.bias"] = _normal((f,), 0.02, gen)
        p[f"{name}.fc2.weight"] = _normal((d, f), in_std, gen)
        p[f"{name}.fc2.bias"] = _normal((d,), 0.02, gen)

    if kind == "vision":
        p["embeddings.class_embedding"] = _normal((d,), d ** -0.5, gen)
        p["embeddings.patch_embedding.weight"] = _normal((d, cfg.num_channels, cfg.patch_size, cfg.patch_size), 0.02, gen)
        p["embeddings.position_embedding.weight"] = _normal((cfg.seq_len, d), 0.02, gen)
        ln("pre_layrnorm")
    else:
        p["embeddings.token_embedding.weight"] = _normal((cfg.vocab_size, d), 0.02, gen)
        p["embeddings.position_embedding.weight"] = _normal((cfg.max_position_embeddings, d), 0.02, gen)
    for i in range(L):
        pre = f"encoder.layers.{i}"
        if cfg.add_time_attn:
            p[pre + ".temporal_embedding"] = _normal((1, cfg.num_frames, d), d ** -0.5, gen)
            attn(pre + ".temporal_attn")
            ln(pre + ".temporal_layer_norm1")
            if cfg.temporal_mlp:
                mlp(pre + ".temporal_mlp")
                ln(pre + ".temporal_layer_norm2")
        attn(pre + ".self_attn")
        ln(pre + ".layer_norm1")
        mlp(pre + ".mlp")
        ln(pre + ".layer_norm2")
    ln("post_layernorm" if kind == "vision" else "final_layer_norm")
    return p


def init_fusion_params(modality_types: Sequence[str], feature_dims: int, fusion_dim: int, num_classes: int,
                       seed: int, head_in: Optional[int] = None, intra_attention: bool = False, dedicated: bool = False,
                       regression: bool = False, distillation: bool = False, inter_attention: bool = False) -> Params:
    """Seeded init for ``modal_sum`` / ``modal_concat`` / ``modal_concat_full`` + ``Head`` (src/model/baseline.py:27-50,66-71)
    parameter names; head_in = width of the fused row (fusion_dim for sum, fusion_dim * M for the concat heads)."""
    gen = torch.Generator().manual_seed(seed)
    head_in = fusion_dim if head_in is None else head_in
    fp: Params = {}
    M = len(modality_types)
    if distillation:                             # modal_distillation (:361-365): modal_proj is a Sequential over the concatenation
        fp["modal_proj.0.weight"] = _normal((fusion_dim, feature_dims * M), (feature_dims * M) ** -0.5, gen)
        fp["modal_proj.0.bias"] = _normal((fusion_dim,), 0.02, gen)
        fp["modal_proj.2.weight"] = _normal((fusion_dim, fusion_dim), fusion_dim ** -0.5, gen)
        fp["modal_proj.2.bias"] = _normal((fusion_dim,), 0.02, gen)
    elif dedicated:                              # modal_dedicated_dnn (:339-343): no modal_proj
        for m in modality_types:
            fp[f"dedicated_dnn.{m}.weight"] = _normal((fusion_dim, feature_dims * (M - 1)), (feature_dims * (M - 1)) ** -0.5, gen)
            fp[f"dedicated_dnn.{m}.bias"] = _normal((fusion_dim,), 0.02, gen)
        fp["dedicated_dnn.full.weight"] = _normal((fusion_dim, feature_dims * M), (feature_dims * M) ** -0.5, gen)
        fp["dedicated_dnn.full.bias"] = _normal((fusion_dim,), 0.02, gen)
    else:
        for m in modality_types:
            fp[f"modal_proj.{m}.weight"] = _normal((fusion_dim, feature_dims), feature_dims ** -0.5, gen)
            fp[f"modal_proj.{m}.bias"] = _normal((fusion_dim,), 0.02, gen)
    fp["norm.weight"] = 1.0 + _normal((head_in,), 0.02, gen)
    fp["norm.bias"] = _normal((head_in,), 0.02, gen)
    fp["head.head.0.weight"] = _normal((fusion_dim, head_in), head_in ** -0.5, gen)
    fp["head.head.0.bias"] = _normal((fusion_dim,), 0.02, gen)
    fp["head.head.3.weight"] = _normal((num_classes, fusion_dim), fusion_dim ** -0.5, gen)
    fp["head.head.3.bias"] = _normal((num_classes,), 0.02, gen)
    if regression:                               # modal_regression (:104-110)
        for s in modality_types:
            for t in modality_types:
                if s != t:
                    fp[f"cross_modal_regressors.{s}_to_{t}.weight"] = _normal((fusion_dim, feature_dims), feature_dims ** -0.5, gen)
                    fp[f"cross_modal_regressors.{s}_to_{t}.bias"] = _normal((fusion_dim,), 0.02, gen)
    if inter_attention:                          # modal_inter_attention (:218-219): query token + nn.MultiheadAttention(D, 4)
        fp["query_token"] = _normal((1, 1, fusion_dim), 1.0, gen)
        fp["attn.in_proj_weight"] = _normal((3 * fusion_dim, fusion_dim), fusion_dim ** -0.5, gen)
        fp["attn.in_proj_bias"] = _normal((3 * fusion_dim,), 0.02, gen)
        fp["attn.out_proj.weight"] = _normal((fusion_dim, fusion_dim), fusion_dim ** -0.5, gen)
        fp["attn.out_proj.bias"] = _normal((fusion_dim,), 0.02, gen)
    if intra_attention:                          # modal_intra_channel_attention (:190-196)
        fp["fusion_representation"] = _normal((1, fusion_dim), 1.0, gen)
        fp["channel_attention.0.weight"] = _normal((fusion_dim // 4, 2 * fusion_dim), (2 * fusion_dim) ** -0.5, gen)
        fp["channel_attention.0.bias"] = _normal((fusion_dim // 4,), 0.02, gen)
        fp["channel_attention.2.weight"] = _normal((fusion_dim, fusion_dim // 4), (fusion_dim // 4) ** -0.5, gen)
        fp["channel_attention.2.bias"] = _normal((fusion_dim,), 0.02, gen)
    return fp


def synth_text_batch(batch: int, ctx: int, seed: int, vocab: int = 49408) -> Tuple[Tensor, Tensor]:
    """Config-2 text recipe: ids uniform in [1000,40000), BOS 49406 first,
    EOS (= pad, the max id) at len-1 and beyond, len ~ U{8..ctx}."""
    gen = torch.Generator().manual_seed(seed)
    bos, eos = vocab - 2, vocab - 1
    lo, hi = (1000, 40000) if vocab > 40000 else (1, max(2, vocab - 2))
    ids = torch.randint(lo, hi, (batch, ctx), generator=gen)
    lens = torch.randint(min(8, ctx), ctx + 1, (batch,), generator=gen)
    pos = torch.arange(ctx)[None]
    ids[:, 0] = bos
    ids = torch.where(pos >= (lens[:, None] - 1), torch.full_like(ids, eos), ids)
    mask = (pos < lens[:, None]).to(torch.int64)
    return ids, mask


def synth_missing_index(batch: int, modality_types: Sequence[str], ratio: float, seed: int,
                        codes: Dict[str, int] = MISSING_TYPE_INDEX) -> Tensor:
    """``simulate_missing_modality`` src/utils/generate_missing.py:8-40, 'mixed' mode restated bit-exactly:
    ``int(n*ratio)`` samples drawn by ``random.sample`` after ``random.seed(seed)``, each given a code
    drawn by ``random.choice`` over the present modalities' codes; all other samples 0."""
    import random
    rng = random.Random(seed)
    pool = [codes[m] for m in modality_types]
    out = [0] * batch
    for idx in rng.sample(range(batch), int(batch * ratio)):
        out[idx] = rng.choice(pool)
    return torch.tensor(out, dtype=torch.int64)
