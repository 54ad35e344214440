"""Import shims that let the reference's own modules run in THIS container (build box only).

TEST INFRASTRUCTURE ONLY (used by oracle/make_golden.py).  /root/reference does not exist on
the GPU box, so nothing at run time may import this module; fixtures made with it are committed
under tests/golden/.

What is shimmed and why (probes P1-P7):
  * ``languagebind`` is entered as a namespace package (so ``languagebind/__init__.py``, which
    imports torchvision/decord/... processors that are absent here, is not executed); the
    ``modeling_*.py`` / ``configuration_*.py`` files themselves are executed unmodified.
  * ``peft`` (absent) gets a stub module; it is only *called* when ``lora_r != 0`` and every
    config built here passes ``lora_r=0``.
  * ``transformers.models.clip.modeling_clip`` 5.x no longer exports ``_expand_mask`` and
    ``clip_loss``, which the reference imports by name (image/modeling_image.py:11-12).
    ``_expand_mask`` is restated from its transformers-4.3x definition; ``clip_loss`` is never
    called on the hot path.
  * ``torch_geometric`` (absent) gets stubs so ``src/model/baseline.py`` imports; the two graph
    fusion heads that need it are out of reach and not exercised.
"""
from __future__ import annotations

import importlib
import sys
import types

import torch

REF_ROOT = "/root/reference"


def _expand_mask(mask: torch.Tensor, dtype: torch.dtype, tgt_len=None):
    bsz, src_len = mask.size()
    tgt_len = tgt_len if tgt_len is not None else src_len
    expanded_mask = mask[:, None, None, :].expand(bsz, 1, tgt_len, src_len).to(dtype)
    inverted_mask = 1.0 - expanded_mask
    return inverted_mask.masked_fill(inverted_mask.to(torch.bool), torch.finfo(dtype).min)


def _clip_loss_unused(*a, **k):  # pragma: no cover
    raise RuntimeError("clip_loss is not part of the hot path")


def install() -> None:
    if getattr(install, "_done", False):
        return
    # 1. peft stub
    if "peft" not in sys.modules:
        peft = types.ModuleType("peft")

        class LoraConfig:  # noqa: D401
            def __init__(self, *a, **k):
                raise RuntimeError("peft stub: lora_r must be 0 in this container")

        def get_peft_model(*a, **k):
            raise RuntimeError("peft stub: lora_r must be 0 in this container")

        peft.LoraConfig = LoraConfig
        peft.get_peft_model = get_peft_model
        sys.modules["peft"] = peft
    # 2. transformers 5.x compat names
    from transformers.models.clip import modeling_clip as mc
    if not hasattr(mc, "_expand_mask"):
        mc._expand_mask = _expand_mask
    if not hasattr(mc, "clip_loss"):
        mc.clip_loss = _clip_loss_unused
    import transformers
    import transformers.utils as tu
    for name in ("add_start_docstrings", "add_start_docstrings_to_model_forward", "replace_return_docstrings"):
        def _passthrough_factory():
            def deco(*a, **k):
                def wrap(fn):
                    return fn
                return wrap
            return deco
        if not hasattr(tu, name):
            setattr(tu, name, _passthrough_factory())
        if not hasattr(transformers, name):
            setattr(transformers, name, getattr(tu, name))
    # 3. languagebind as a namespace package rooted at the reference tree
    if "languagebind" not in sys.modules:
        pkg = types.ModuleType("languagebind")
        pkg.__path__ = [REF_ROOT + "/languagebind"]
        sys.modules["languagebind"] = pkg
        for sub in ("image", "video", "audio", "depth", "thermal"):
            sp = types.ModuleType(f"languagebind.{sub}")
            sp.__path__ = [f"{REF_ROOT}/languagebind/{sub}"]
            sys.modules[f"languagebind.{sub}"] = sp
    # 4. torch_geometric stubs (graph heads unreachable)
    if "torch_geometric" not in sys.modules:
        tg = types.ModuleType("torch_geometric")
        tgnn = types.ModuleType("torch_geometric.nn")
        tgdata = types.ModuleType("torch_geometric.data")

        class _Missing:
            def __init__(self, *a, **k):
                raise RuntimeError("torch_geometric is absent in this container")

        tgnn.SuperGATConv = _Missing
        tgdata.Batch = _Missing
        tgdata.Data = _Missing
        tg.nn, tg.data = tgnn, tgdata
        sys.modules.update({"torch_geometric": tg, "torch_geometric.nn": tgnn, "torch_geometric.data": tgdata})
    install._done = True


def ref_modeling(modality: str):
    """Return (modeling module, configuration module) of the reference for one modality."""
    install()
    mod = importlib.import_module(f"languagebind.{modality}.modeling_{modality}")
    cfg = importlib.import_module(f"languagebind.{modality}.configuration_{modality}")
    return mod, cfg


def ref_baseline():
    """Return the reference's ``src.model.baseline`` module (needs the languagebind namespace to
    export the names it imports; they are never used by the fusion heads)."""
    install()
    lb = sys.modules["languagebind"]
    for name in ("LanguageBind", "to_device", "transform_dict", "LanguageBindImageTokenizer"):
        if not hasattr(lb, name):
            setattr(lb, name, None)
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    return importlib.import_module("src.model.baseline")


def ref_generate_missing():
    install()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    return importlib.import_module("src.utils.generate_missing")
