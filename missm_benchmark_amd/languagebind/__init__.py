"""Drop-in for the reference's ``languagebind`` package surface used by the hot path
(reference: languagebind/__init__.py:32-89).

``LanguageBind(clip_type, use_temp=True, cache_dir=...)`` keeps the reference signature, attributes
(``modality_encoder`` / ``modality_proj`` ModuleDicts keyed by modality plus ``'language'``, the un-registered
``modality_scale`` dict, ``modality_config``) and ``forward(inputs) -> {modality: [B, projection_dim]}``.

Differences that are build decisions (SURVEY.md sections 0.2, 8b):
  * checkpoints cannot be fetched by NAME here (no network): ``from_pretrained`` resolves a local path, ``<cache_dir>/<name>``
    or the hub cache layout holding ``config.json`` + a state dict and fails loudly otherwise (like the reference); the
    synthetic ViT-B/16-class config below with a seeded init (what the benchmark uses) is built on request only:
    ``configs=`` or ``allow_synthetic=True``;
  * towers run on HIP kernels in ``compute_dtype`` (bf16 default, fp32 for parity) - see ``towers.ClipTower``;
  * image/video/audio preprocessing and the BPE tokenizer are outside this path.
"""
from __future__ import annotations

import json
import math
import os
from dataclasses import asdict
from typing import Dict, Optional

import torch
from torch import nn

from .. import nn as hnn
from ..towers import ClipTower, TowerConfig, pooled_output_only, run_towers

LOGIT_SCALE_INIT = 2.6592  # configuration_image.py:303
PROJECTION_DIM = 768       # train_ddp.py:31 feature_dims forces this


def default_vision_config(modality: str) -> TowerConfig:
    """Synthetic ViT-B/16 config of SURVEY.md section 0.2 (video: factorised time attention over 8 frames)."""
    if modality == "video":
        return TowerConfig(kind="vision", add_time_attn=True, num_frames=8)
    return TowerConfig(kind="vision")


def default_text_config() -> TowerConfig:
    return TowerConfig(kind="text")


class _Projection(nn.Module):
    """``nn.Linear(hidden, projection_dim, bias=False)`` (image/modeling_image.py:766-767) on the HIP small-linear kernel."""

    def __init__(self, in_features: int, out_features: int, seed: int):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.weight = nn.Parameter(torch.randn(out_features, in_features, generator=g) * in_features ** -0.5)
        self.in_features, self.out_features = in_features, out_features

    def forward(self, x):
        return hnn._LinearFn.apply(x, self.weight, None, False, None, 0)


def merge_lora_state_dict(sd: Dict[str, torch.Tensor], lora_r: int, lora_alpha: float) -> Dict[str, torch.Tensor]:
    """Released LanguageBind checkpoints carry the encoder wrapped by peft (reference image/modeling_image.py:775-793:
    ``get_peft_model(vision_model.encoder, LoraConfig(r, lora_alpha, target_modules=q/k/v/out_proj ...))``), i.e. keys
    ``...encoder.base_model.model.layers.N.self_attn.q_proj.{base_layer.weight, lora_A.default.weight, lora_B.default.weight}``.
    The towers here keep plain weights: fold every adapter into its base matrix, W + (lora_alpha / r) * B @ A - the same
    function of the input as the unmerged peft forward (lora_dropout is inactive in eval) - and strip the wrapper prefixes.
    Host-side, once, at load time."""
    scale = float(lora_alpha) / float(lora_r) if lora_r else 0.0
    out: Dict[str, torch.Tensor] = {}
    lora = {}
    for k, v in sd.items():
        k = k.replace(".base_model.model.", ".")
        if ".lora_A." in k or ".lora_B." in k:
            stem, which = (k.split(".lora_A.")[0], "A") if ".lora_A." in k else (k.split(".lora_B.")[0], "B")
            lora.setdefault(stem, {})[which] = v
            continue
        if ".lora_dropout." in k or ".lora_embedding_" in k:
            continue
        out[k.replace(".base_layer.", ".")] = v
    for stem, ab in lora.items():
        if "A" not in ab or "B" not in ab or stem + ".weight" not in out:
            raise KeyError(f"incomplete LoRA adapter for {stem}")
        out[stem + ".weight"] = out[stem + ".weight"].float() + scale * (ab["B"].float() @ ab["A"].float())
    return out


def resize_pos_embed(weight: torch.Tensor, grid, extra_tokens: int = 1) -> torch.Tensor:
    """``resize_pos`` of the reference (image/modeling_image.py:795-839; the audio model maps its square position grid onto the
    (num_mel_bins, target_length) spectrogram): the class-token rows are kept, the patch rows - a square grid - are resampled to
    ``grid`` = (rows, columns) with antialiased bicubic interpolation (``F.interpolate(..., mode='bicubic', antialias=True,
    align_corners=False)``, the reference's own call).  Host-side, once, at load time, like the LoRA merge."""
    import torch.nn.functional as F
    gh, gw = int(grid[0]), int(grid[1])
    if gh * gw + extra_tokens == weight.shape[0]:
        return weight
    tok, img = weight[:extra_tokens], weight[extra_tokens:]
    old = int(math.sqrt(img.shape[0]))
    if old * old != img.shape[0]:
        raise ValueError(f"position embedding of {img.shape[0]} patch rows is not a square grid")
    img = img.float().reshape(1, old, old, -1).permute(0, 3, 1, 2)
    img = F.interpolate(img, size=(gh, gw), mode="bicubic", antialias=True, align_corners=False)
    img = img.permute(0, 2, 3, 1).reshape(gh * gw, -1)
    return torch.cat([tok.float(), img], dim=0).to(weight.dtype)


def _vision_config_from_json(raw: dict, modality: str = "image") -> TowerConfig:
    fields = TowerConfig.__dataclass_fields__
    kw = {k: v for k, v in raw.items() if k in fields and k != "kind"}
    if modality != "video" and kw.get("add_time_attn"):
        kw["temporal_mlp"] = True       # the image-family layer keeps the time branch's MLP (image/modeling_image.py:83-84), the video file dropped it
    if raw.get("num_mel_bins", 0) and raw.get("target_length", 0):      # audio (image/modeling_image.py:797-798): spectrogram image
        kw["image_size"] = (int(raw["num_mel_bins"]), int(raw["target_length"]))
    elif isinstance(kw.get("image_size"), list):
        kw["image_size"] = tuple(kw["image_size"])
    return TowerConfig(kind="vision", **kw)


class LanguageBindModel(nn.Module):
    """One modality's CLIP pair (reference ``LanguageBind<Modality>``, image/modeling_image.py:734-768): vision tower,
    text tower, the two bias-free projections and ``logit_scale``."""

    modality = "image"

    def __init__(self, vision_config: Optional[TowerConfig] = None, text_config: Optional[TowerConfig] = None,
                 projection_dim: int = PROJECTION_DIM, logit_scale_init_value: float = LOGIT_SCALE_INIT,
                 compute_dtype: torch.dtype = torch.bfloat16, seed: int = 0, build_text: bool = True):
        super().__init__()
        vc = vision_config or default_vision_config(self.modality)
        tc = text_config or default_text_config()
        if vc.kind != "vision" or tc.kind != "text":
            raise ValueError("config.vision_config / config.text_config are of the wrong kind")
        self.config = {"vision_config": asdict(vc), "text_config": asdict(tc), "projection_dim": projection_dim,
                       "logit_scale_init_value": logit_scale_init_value}
        self.vision_model = ClipTower(vc, compute_dtype, seed=seed)
        self.visual_projection = _Projection(vc.hidden_size, projection_dim, seed + 1)
        if build_text:
            self.text_model = ClipTower(tc, compute_dtype, seed=seed + 2)
            self.text_projection = _Projection(tc.hidden_size, projection_dim, seed + 3)
        self.logit_scale = nn.Parameter(torch.tensor(float(logit_scale_init_value)))

    @classmethod
    def resolve_checkpoint_dir(cls, pretrained_model_name_or_path: str, cache_dir: Optional[str] = None) -> Optional[str]:
        """Local-only resolution (there is no network): the path itself, ``<cache_dir>/<name>``, or the Hugging Face cache
        layout ``<cache_dir>/models--<org>--<name>/snapshots/<rev>/`` - whichever holds a ``config.json``."""
        name = pretrained_model_name_or_path
        roots = [name, os.path.join(cache_dir or ".", name)]
        hub = os.path.join(cache_dir or ".", "models--" + name.replace("/", "--"), "snapshots")
        if os.path.isdir(hub):
            roots += sorted(os.path.join(hub, r) for r in os.listdir(hub))
        for root in roots:
            if os.path.isdir(root) and os.path.exists(os.path.join(root, "config.json")):
                return root
        return None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, cache_dir: Optional[str] = None, *, allow_synthetic: bool = False,
                        merge_lora: bool = False, **kw):
        """A directory with ``config.json`` and ``pytorch_model.bin`` / ``model.pt`` / ``model.pth`` (see
        ``resolve_checkpoint_dir``).  Like the reference (languagebind/__init__.py:63-64) this FAILS when the checkpoint
        cannot be found or is incomplete; a seeded synthetic model is built only on request (``allow_synthetic=True``, or
        ``LanguageBind(configs=...)``).
        A peft-wrapped checkpoint (``vision_config.lora_r`` > 0, adapter keys in the state dict) loads as the reference builds it
        (image/modeling_image.py:775-793): frozen encoder + trainable rank-r adapters, fine-tuned adapter-only.  ``merge_lora=True``
        folds the adapters into plain weights instead (full fine-tuning of the merged model, the round-2 behaviour)."""
        root = cls.resolve_checkpoint_dir(pretrained_model_name_or_path, cache_dir)
        if root is None:
            if allow_synthetic:
                return cls(**kw)
            raise FileNotFoundError(f"checkpoint {pretrained_model_name_or_path!r} not found (looked in it, under cache_dir="
                                    f"{cache_dir!r} and in its hub layout); there is no network here - pass a local directory, "
                                    "LanguageBind(configs=...) or allow_synthetic=True for seeded random towers")
        raw = json.load(open(os.path.join(root, "config.json")))
        fields = TowerConfig.__dataclass_fields__
        files = [os.path.join(root, fn) for fn in ("pytorch_model.bin", "model.pt", "model.pth") if os.path.exists(os.path.join(root, fn))]
        if not files:
            raise FileNotFoundError(f"{root} holds a config.json but no pytorch_model.bin / model.pt / model.pth")
        sd = torch.load(files[0], map_location="cpu")
        vraw = dict(raw.get("vision_config", {}))
        if any(".lora_A." in k for k in sd):
            if merge_lora:
                sd = merge_lora_state_dict(sd, int(vraw.get("lora_r", 2)), float(vraw.get("lora_alpha", 16)))
                vraw["lora_r"] = 0
            else:
                vraw.setdefault("lora_r", 2)       # configuration_image.py:200-201 defaults
                vraw.setdefault("lora_alpha", 16)
        else:
            vraw["lora_r"] = 0                  # plain weights: nothing was wrapped when this checkpoint was written
        vc = _vision_config_from_json(vraw, cls.modality)
        tc = TowerConfig(kind="text", **{k: v for k, v in raw.get("text_config", {}).items() if k in fields and k != "kind"})
        kw.pop("text_config", None)
        kw.pop("projection_dim", None)
        model = cls(vc, tc, raw.get("projection_dim", PROJECTION_DIM), raw.get("logit_scale_init_value", LOGIT_SCALE_INIT), **kw)
        sd = {k: v for k, v in sd.items() if not k.endswith("position_ids")}     # non-persistent buffers in newer layouts
        pk = "vision_model.embeddings.position_embedding.weight"
        if pk in sd and sd[pk].shape[0] != vc.seq_len:                         # a checkpoint trained on another patch grid
            sd[pk] = resize_pos_embed(sd[pk], vc.grid)
        res = model.load_state_dict(sd, strict=False)
        has_text = hasattr(model, "text_model")
        bad = [k for k in res.unexpected_keys if has_text or not k.startswith(("text_model.", "text_projection."))]
        if res.missing_keys or bad:
            raise KeyError(f"{files[0]} does not match the model built from {root}/config.json: missing {res.missing_keys[:8]}"
                           f"{' ...' if len(res.missing_keys) > 8 else ''}, unexpected {bad[:8]}{' ...' if len(bad) > 8 else ''}")
        return model


def _model_class(modality: str):
    return type(f"LanguageBind{modality.capitalize()}", (LanguageBindModel,), {"modality": modality})


LanguageBindImage, LanguageBindVideo, LanguageBindDepth, LanguageBindAudio, LanguageBindThermal = (
    _model_class(m) for m in ("image", "video", "depth", "audio", "thermal"))

model_dict = {"thermal": LanguageBindThermal, "image": LanguageBindImage, "video": LanguageBindVideo,
              "depth": LanguageBindDepth, "audio": LanguageBindAudio}
config_dict = {m: TowerConfig for m in model_dict}


class _OutOfScope:
    def __init__(self, what):
        self.what = what

    def __call__(self, *a, **k):
        raise NotImplementedError(f"{self.what} is host-side preprocessing outside the MI355X hot path (SURVEY.md 2.1); "
                                  "feed tensors (pixel_values / input_ids) directly")

    from_pretrained = __call__


def _gpu_transform(kind):
    def make(config=None, **kw):
        from .. import processing
        if kind == "depth":
            vc = (config or {}).get("vision_config", {}) if isinstance(config, dict) else {}
            return processing.DepthTransform(max_depth=float(vc.get("max_depth", 10)), **kw)
        if kind == "audio":
            return processing.AudioTransform(config, **kw)
        return processing.ImageTransform(**kw)
    return make


# image / thermal / depth / audio: GPU-side transforms (processing.py); video decoding (decord / cv2 / pytorchvideo) is a host
# library stack that is not in this image - that entry raises
transform_dict = {m: _OutOfScope(f"{m} processor") for m in model_dict}
transform_dict.update({"image": _gpu_transform("image"), "thermal": _gpu_transform("thermal"), "depth": _gpu_transform("depth"),
                       "audio": _gpu_transform("audio")})
LanguageBindImageTokenizer = _OutOfScope("LanguageBindImageTokenizer")


import os as _os
# scheduling experiments: MISSM_STREAM_GROUPS="image=a,audio=a,depth=b,thermal=b" shares streams; MISSM_STREAM_PRIO="video=-1"
_STREAM_GROUP = dict(kv.split("=") for kv in _os.environ.get("MISSM_STREAM_GROUPS", "").split(",") if "=" in kv)
_STREAM_PRIO = {k: int(v) for k, v in (kv.split("=") for kv in _os.environ.get("MISSM_STREAM_PRIO", "").split(",") if "=" in kv)}


class LanguageBind(nn.Module):
    def __init__(self, clip_type, use_temp=True, cache_dir="./cache_dir", *, configs: Optional[Dict[str, TowerConfig]] = None,
                 text_config: Optional[TowerConfig] = None, projection_dim: int = PROJECTION_DIM,
                 compute_dtype: torch.dtype = torch.bfloat16, seed: int = 0, allow_synthetic: bool = False):
        super().__init__()
        self.use_temp = use_temp
        self.parallel_streams = True
        self.group_towers = _os.environ.get("MISSM_TOWER_GROUPS", "1") != "0"   # lock-step + grouped GEMMs for shape-identical towers
        self._streams = {}
        self._scale_cache = {}
        encoders, projs = {}, {}
        self.modality_scale = {}   # plain dict on purpose: un-registered in the reference too (languagebind/__init__.py:60,67)
        self.modality_config = {}
        items = list(clip_type.items())
        model = None
        for i, (k, v) in enumerate(items):
            last = i == len(items) - 1
            kw = dict(compute_dtype=compute_dtype, seed=seed + 10 * i, build_text=last)
            if configs and k in configs:
                model = model_dict[k](configs[k], text_config, projection_dim, **kw)
            else:
                model = model_dict[k].from_pretrained(f"LanguageBind/{v}", cache_dir=cache_dir, text_config=text_config,
                                                      projection_dim=projection_dim, allow_synthetic=allow_synthetic, **kw)
            encoders[k] = model.vision_model
            projs[k] = model.visual_projection
            self.modality_scale[k] = model.logit_scale
            self.modality_config[k] = model.config
        # the text tower always comes from the LAST loaded modality model (languagebind/__init__.py:69-70)
        encoders["language"] = model.text_model
        projs["language"] = model.text_projection
        self.modality_encoder = nn.ModuleDict(encoders)
        self.modality_proj = nn.ModuleDict(projs)

    def set_compute_dtype(self, dtype: torch.dtype):
        for t in self.modality_encoder.values():
            t.set_compute_dtype(dtype)
        return self

    def _finish(self, key, pooled):
        emb = self.modality_proj[key](pooled)
        scale = 1.0
        if self.use_temp and key != "language":
            # exp(logit_scale) is a launch argument: read the (un-registered, normally CPU-resident) parameter only when it
            # changed - on a GPU-resident one this would otherwise be a device-to-host stall per tower per forward
            ls = self.modality_scale[key]
            cached = self._scale_cache.get(key)
            if cached is None or cached[0] != ls._version or cached[1] is not ls:
                cached = self._scale_cache[key] = (ls._version, ls, math.exp(float(ls.detach())))
            scale = cached[2]
        return hnn.l2norm_scale(emb, scale)

    def _embed(self, key, value):
        with pooled_output_only():       # (only [1], the pooled output, is read: languagebind/__init__.py:78)
            return self._finish(key, self.modality_encoder[key](**value)[1])

    def _embed_unit(self, keys, values):
        """one work unit: a single tower, or several shape-identical ones in lock-step (towers.run_towers)"""
        if len(keys) == 1:
            return {keys[0]: self._embed(keys[0], values[0])}
        with pooled_output_only():
            outs = run_towers([self.modality_encoder[k] for k in keys], values)
        return {k: self._finish(k, o[1]) for k, o in zip(keys, outs)}

    def _units(self, inputs):
        """[(keys, values)]: modalities whose towers share config, compute dtype and input shape form one unit (their linears are
        launched as grouped GEMMs: four B x 197-row towers fill the chip like one long tower); most expensive unit first"""
        cost = lambda kv: -sum(v.numel() for v in kv[1].values() if torch.is_tensor(v))   # noqa: E731
        order = sorted(inputs.items(), key=cost)
        units, by_sig = [], {}
        for key, value in order:
            t = self.modality_encoder[key]
            px = value.get("pixel_values") if isinstance(value, dict) else None
            sig = None
            if self.group_towers and key != "language" and torch.is_tensor(px) and isinstance(t, ClipTower):
                sig = (repr(t.config), t.compute_dtype, tuple(px.shape))
            if sig is not None and sig in by_sig:
                by_sig[sig][0].append(key); by_sig[sig][1].append(value)
                continue
            unit = ([key], [value])
            units.append(unit)
            if sig is not None:
                by_sig[sig] = unit
        return units

    def forward(self, inputs):
        """Same contract as the reference loop (languagebind/__init__.py:75-85).  The reference encodes the modalities one
        after the other on one stream; here shape-identical towers run in lock-step as one unit, and the units are spread over
        two HIP streams.  autograd replays each unit's backward on the stream its forward ran on."""
        dev = next(iter(self.modality_proj.values())).weight.device
        units = self._units(inputs)
        done = {}
        if dev.type != "cuda" or not self.parallel_streams or len(units) < 2:
            for keys, values in units:
                done.update(self._embed_unit(keys, values))
            return {key: done[key] for key in inputs}
        main = torch.cuda.current_stream()
        # Enqueue the most expensive unit first (video: T x the tokens of an image-like tower).  Its forward then overlaps the
        # launches of the others from the start, and - autograd replays later-created nodes first - its backward runs last,
        # alone, with full grids (measured: 358 -> 368 samples/s at B = 32, 5 modalities).  Results do not depend on the order.
        # Two streams: the most expensive unit on its own, all the others one after the other on a second one (measured at
        # B = 32, video + 4 image-like towers: one stream per tower 371, two shared streams for the four 372, one 379 samples/s;
        # raising the video stream's priority 364)
        used = []
        for i, (keys, values) in enumerate(units):
            skey = _STREAM_GROUP.get(keys[0], ("_big" if i == 0 else "_rest") if not _STREAM_GROUP else keys[0])
            st = self._streams.get(skey)
            if st is None:
                st = self._streams[skey] = torch.cuda.Stream(device=dev, priority=_STREAM_PRIO.get(keys[0], 0))
            used.append(st)
            st.wait_stream(main)
            with torch.cuda.stream(st):
                outs = self._embed_unit(keys, values)
            for out in outs.values():
                out.record_stream(main)
            done.update(outs)
        for st in used:
            main.wait_stream(st)
        return {key: done[key] for key in inputs}      # the reference's (input) order


def to_device(x, device):
    """reference languagebind/__init__.py:87-89; host tensors travel through pinned memory with an asynchronous copy"""
    from ..processing import to_device_async
    return to_device_async(x, device)
