"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports every symbol the header declares."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from missm_benchmark_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "missm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(missm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/missm_hip.h but not exported"


def test_binding_covers_header():
    from missm_benchmark_amd import _lib
    bound = set(_lib.SIGNATURES) | set(_lib.PLAIN)
    assert bound == set(_header_symbols())


def test_argument_counts_match_header():
    from missm_benchmark_amd import _lib
    text = open(os.path.join(ROOT, "include", "missm_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, args in _lib.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^)]*)\)", text)
        assert m, name
        assert len([a for a in m.group(1).split(",") if a.strip()]) == len(args), name


def test_error_channel_without_gpu(lib):
    # argument validation happens on the host before any launch, so it is testable without a GPU
    rc = lib.missm_gemm_nt(None, None, None, 0, 0, 0, 0, 0, 0, 1.0, None, None, None, None, 0, 0, 0, 0, 1, None)
    assert rc == -1
    assert b"gemm" in lib.missm_last_error()
    from missm_benchmark_amd import _lib as L
    assert lib.missm_abi_version() == L.ABI_VERSION >= 3
    assert lib.missm_device_count() >= 0


def test_product_path_has_no_oracle_import():
    pkg = os.path.join(ROOT, "missm_benchmark_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "missm_oracle" not in src and "ref_shims" not in src, f"{f} touches the oracle"


def test_lora_checkpoint_merge_matches_unmerged_adapter(tmp_path):
    """A peft-wrapped reference checkpoint (image/modeling_image.py:775-793) loads through from_pretrained(local dir): adapter
    keys are folded into the base weights, W + alpha/r * B A, which is the same function as the unmerged forward."""
    import json
    import torch
    from missm_benchmark_amd.languagebind import LanguageBindImage, merge_lora_state_dict
    from missm_benchmark_amd.towers import TowerConfig
    from dataclasses import asdict
    tiny = dict(hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=2, image_size=32, patch_size=16)
    vc, tc = TowerConfig(kind="vision", **tiny), TowerConfig(kind="text", hidden_size=32, intermediate_size=64, num_hidden_layers=1,
                                                              num_attention_heads=2, vocab_size=64, max_position_embeddings=8)
    src = LanguageBindImage(vc, tc, projection_dim=16, seed=3)
    plain = {k: v.clone() for k, v in src.state_dict().items()}
    g = torch.Generator().manual_seed(0)
    r, alpha = 2, 16
    wrapped, adapters = {}, {}
    for k, v in plain.items():
        if k.startswith("vision_model.encoder."):
            rest = k[len("vision_model.encoder."):]
            if any(f"self_attn.{n}_proj." in rest for n in ("q", "k", "v", "out")):
                wrapped["vision_model.encoder.base_model.model." + rest.replace("_proj.", "_proj.base_layer.")] = v
                if rest.endswith(".weight"):
                    stem = "vision_model.encoder.base_model.model." + rest[:-len(".weight")]
                    A, Bm = torch.randn(r, v.shape[1], generator=g) * 0.1, torch.randn(v.shape[0], r, generator=g) * 0.1
                    wrapped[stem + ".lora_A.default.weight"], wrapped[stem + ".lora_B.default.weight"] = A, Bm
                    adapters[k] = (A, Bm)
            else:
                wrapped["vision_model.encoder.base_model.model." + rest] = v
        else:
            wrapped[k] = v
    merged = merge_lora_state_dict(wrapped, r, alpha)
    assert set(merged) == set(plain)
    x = torch.randn(5, 32, generator=g)
    for k, (A, Bm) in adapters.items():
        unmerged = x @ plain[k].t() + (alpha / r) * (x @ A.t()) @ Bm.t()         # peft's forward
        assert torch.allclose(x @ merged[k].t(), unmerged, atol=1e-5)
    untouched = [k for k in plain if k not in adapters]
    assert all(torch.equal(merged[k], plain[k]) for k in untouched)
    d = tmp_path / "LanguageBind" / "LanguageBind_Image"
    d.mkdir(parents=True)
    json.dump({"vision_config": dict(asdict(vc), lora_r=r, lora_alpha=alpha), "text_config": asdict(tc), "projection_dim": 16}, open(d / "config.json", "w"))
    torch.save(wrapped, d / "pytorch_model.bin")
    loaded = LanguageBindImage.from_pretrained("LanguageBind/LanguageBind_Image", cache_dir=str(tmp_path), seed=9, merge_lora=True)
    for k, v in loaded.state_dict().items():
        assert torch.allclose(v, merged[k], atol=1e-6), k
    # default: as the reference builds it - the peft-wrapped keys load one to one into a LoRA tower (frozen encoder, trainable adapters)
    lora = LanguageBindImage.from_pretrained("LanguageBind/LanguageBind_Image", cache_dir=str(tmp_path), seed=9)
    assert lora.vision_model.lora and lora.vision_model.config.lora_r == r
    sd = lora.state_dict()
    assert set(sd) == set(wrapped)
    for k, v in wrapped.items():
        assert torch.equal(sd[k], v), k
    train = {k for k, p in lora.vision_model.named_parameters() if p.requires_grad}
    assert all((".lora_" in k) or not k.startswith("encoder.") for k in train) and any(".lora_A." in k for k in train)
    assert {"embeddings.class_embedding", "pre_layrnorm.weight", "post_layernorm.bias"} <= train
    assert not any(p.requires_grad for k, p in lora.vision_model.named_parameters() if k.startswith("encoder.") and ".lora_" not in k)


def test_config_options_are_validated_and_patch_dropout_draws_like_the_reference():
    import pytest
    import torch
    from missm_benchmark_amd.towers import ClipTower, PatchDropout, TowerConfig
    tiny = dict(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2, image_size=32, patch_size=16)
    with pytest.raises(ValueError, match="force_patch_dropout"):
        ClipTower(TowerConfig(kind="vision", force_patch_dropout=1.0, **tiny))
    with pytest.raises(ValueError, match="temporal_mlp"):
        ClipTower(TowerConfig(kind="vision", temporal_mlp=True, **tiny))             # the time branch's MLP needs add_time_attn
    t = ClipTower(TowerConfig(kind="vision", force_patch_dropout=0.5, **tiny))
    assert t.patch_dropout.prob == 0.5 and t.patch_dropout.num_keep(196) == 98 and PatchDropout(0.99).num_keep(4) == 1
    # the draw is the reference's (image/modeling_image.py:47-56): randn(batch, tokens).topk(k) on the CPU generator
    torch.manual_seed(5)
    a = t.patch_dropout.keep_indices(3, 1, 16)
    torch.manual_seed(5)
    assert torch.equal(a, torch.randn(3, 16).topk(8, dim=-1).indices)
    torch.manual_seed(6)
    b = t.patch_dropout.keep_indices(2, 4, 16)                                         # T > 1: one draw per SAMPLE
    torch.manual_seed(6)
    assert torch.equal(b, torch.randn(2, 16).topk(8, dim=-1).indices)
    tm = ClipTower(TowerConfig(kind="vision", add_time_attn=True, num_frames=2, temporal_mlp=True, **tiny))
    keys = set(tm.state_dict())
    assert {"encoder.layers.0.temporal_mlp.fc1.weight", "encoder.layers.0.temporal_mlp.fc2.bias",
            "encoder.layers.0.temporal_layer_norm2.weight"} <= keys
