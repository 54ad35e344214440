"""Generate tests/golden/*.pt by RUNNING THE REFERENCE (build container only).

TEST INFRASTRUCTURE ONLY.  Usage (from the repo root, in the build container where
/root/reference is mounted):

    python oracle/make_golden.py

Each fixture holds seeded inputs, the parameters loaded into the reference module, and the
outputs (and selected gradients) the reference's own code produced.  The reference's source
never enters the repo; only these tensors do.  Reference entry points used:

  vision_tiny / video_tiny : languagebind/{image,video}/modeling_*.py  CLIPVisionTransformer.forward
  text_tiny                : stock transformers CLIPTextModel built from a local config (the third-party
                             arithmetic the reference imports; the reference's own CLIPTextTransformer
                             drops its causal mask under transformers 5.x - SURVEY.md P7) and, as a
                             second witness, the reference's CLIPEncoder driven with the merged mask
  fusion_sum               : src/model/baseline.py  finetune_model / modal_sum / Head
  missing_index            : src/utils/generate_missing.py  simulate_missing_modality
  vitb16_config1           : reference image tower, ViT-B/16, B=4 (BASELINE.json configs[0]); weights are
                             NOT stored (343 MB) - they come from oracle.init_tower_params(cfg, seed=0)
"""
from __future__ import annotations

import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import missm_oracle as O  # noqa: E402
import ref_shims  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.manual_seed(0)
torch.set_grad_enabled(True)


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def _grads(module, names):
    sd = dict(module.named_parameters())
    return {n: sd[n].grad.detach().clone() for n in names}


def vision_fixture(name, modality, cfg: O.VisionCfg, batch, seed_w, seed_x, store_params=True, grad_names=(), compact=False):
    mod, cfgm = ref_shims.ref_modeling(modality)
    vc = cfgm.CLIPVisionConfig(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                               num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                               image_size=cfg.image_size, patch_size=cfg.patch_size, lora_r=0,
                               add_time_attn=cfg.add_time_attn, num_frames=cfg.num_frames,
                               force_patch_dropout=getattr(cfg, "force_patch_dropout", 0.0))
    tower = mod.CLIPVisionTransformer(vc).eval()
    params = O.init_tower_params(cfg, seed_w)
    res = tower.load_state_dict(params, strict=False)
    assert not res.unexpected_keys and all("position_ids" in k for k in res.missing_keys), res
    if cfg.num_frames > 1:
        x = torch.randn(batch, cfg.num_channels, cfg.num_frames, cfg.image_size, cfg.image_size, generator=_gen(seed_x))
    else:
        x = torch.randn(batch, cfg.num_channels, cfg.image_size, cfg.image_size, generator=_gen(seed_x))
    keep = None
    if getattr(cfg, "force_patch_dropout", 0.0) > 0:
        # PatchDropout acts in training mode: the reference draws torch.randn(batch, num_tokens).topk(k) from the global CPU generator
        # inside forward (its first RNG use: attention dropout is 0).  Seed, run, then re-seed and repeat the draw to learn the indices.
        tower.train()
        torch.manual_seed(seed_x + 500)
        last, pooled = tower(x, return_dict=False)[:2]
        torch.manual_seed(seed_x + 500)
        ntok = cfg.num_patches
        kk = max(1, int(ntok * (1 - cfg.force_patch_dropout)))
        keep = torch.randn(batch if cfg.num_frames > 1 else x.shape[0], ntok).topk(kk, dim=-1).indices
    else:
        last, pooled = tower(x, return_dict=False)[:2]
    fix = {"cfg": cfg.__dict__.copy(), "seed_w": seed_w, "pixel_values": x, "last_hidden_state": last.detach(),
           "pooled": pooled.detach()}
    if keep is not None:
        fix["patch_keep"], fix["patch_seed"] = keep, seed_x + 500
    if grad_names:
        cot_p = torch.randn(pooled.shape, generator=_gen(seed_x + 100))
        cot_h = torch.randn(last.shape, generator=_gen(seed_x + 101)) * 0.1
        ((pooled * cot_p).sum() + (last * cot_h).sum()).backward()
        fix.update(cot_pooled=cot_p, cot_last=cot_h, grads=_grads(tower, grad_names))
    if store_params:
        fix["params"] = params
    if compact:  # big shapes: inputs/cotangents are regenerated from their seeds (torch.randn, Generator(seed_x[+100,+101]))
        fix["seed_x"], fix["batch"] = seed_x, batch
        fix["last_hidden_slice"] = fix.pop("last_hidden_state")[:, :4, :64].clone()
        for k in ("pixel_values", "cot_pooled", "cot_last"):
            fix.pop(k, None)
        if "grads" in fix:  # keep a 64x64 corner / first 64 entries of each captured gradient
            fix["grads"] = {k: (v[:64, :64] if v.dim() == 2 else v[:64]).clone() for k, v in fix["grads"].items()}
    torch.save(fix, os.path.join(OUT, name + ".pt"))
    # cross-check the oracle right away
    with torch.no_grad():
        h, p = O.vision_tower(x, params, cfg, patch_keep=keep)
    print(f"{name}: ref-vs-oracle last {float((h - last).abs().max()):.2e} pooled {float((p - pooled).abs().max()):.2e}")


def text_fixture(name, cfg: O.TextCfg, batch, seed_w, seed_x, grad_names=(), compact=False):
    from transformers import CLIPTextConfig, CLIPTextModel
    tc = CLIPTextConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                        max_position_embeddings=cfg.max_position_embeddings, hidden_act=cfg.hidden_act,
                        eos_token_id=cfg.vocab_size - 1, bos_token_id=cfg.vocab_size - 2, pad_token_id=cfg.vocab_size - 1,
                        attn_implementation="eager")
    stock = CLIPTextModel(tc).eval()
    params = O.init_tower_params(cfg, seed_w, kind="text")
    core = getattr(stock, "text_model", stock)  # transformers 5.x flattened CLIPTextModel
    res = core.load_state_dict(params, strict=False)
    assert not res.unexpected_keys and all("position_ids" in k for k in res.missing_keys), res
    ids, mask = O.synth_text_batch(batch, cfg.max_position_embeddings, seed_x, vocab=cfg.vocab_size)
    out = stock(input_ids=ids, attention_mask=mask)
    last, pooled = out.last_hidden_state, out.pooler_output
    fix = {"cfg": cfg.__dict__.copy(), "seed_w": seed_w, "input_ids": ids, "attention_mask": mask, "params": params,
           "last_hidden_state": last.detach(), "pooled": pooled.detach()}
    if grad_names:
        cot_p = torch.randn(pooled.shape, generator=_gen(seed_x + 100))
        (pooled * cot_p).sum().backward()
        sd = dict(core.named_parameters())
        fix.update(cot_pooled=cot_p, grads={n: sd[n].grad.detach().clone() for n in grad_names})
    # second witness: the reference's own CLIPEncoder / CLIPEncoderLayer code with causal+padding merged
    mod, cfgm = ref_shims.ref_modeling("image")
    rtc = cfgm.CLIPTextConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                              num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                              max_position_embeddings=cfg.max_position_embeddings)
    rt = mod.CLIPTextTransformer(rtc).eval()
    rt.load_state_dict(params, strict=False)
    with torch.no_grad():
        h0 = rt.embeddings(input_ids=ids)
        merged = mod._make_causal_mask(ids.shape, h0.dtype, device=h0.device) + ref_shims._expand_mask(mask, h0.dtype)
        hr = rt.final_layer_norm(rt.encoder(inputs_embeds=h0, attention_mask=merged, return_dict=False)[0])
        pr = hr[torch.arange(batch), ids.to(torch.int).argmax(-1)]
    fix["pooled_ref_encoder"] = pr
    if compact:  # full-size tower: weights by recipe (seed_w), slices of the big tensors only
        fix.pop("params")
        fix["last_hidden_slice"] = fix.pop("last_hidden_state")[:, :, :64].clone()
        g = fix.get("grads", {})
        tok = "embeddings.token_embedding.weight"
        if tok in g:   # rows that actually receive gradient: the ids of the batch (BOS, EOS, a few words)
            rows = torch.unique(ids)[:48]
            fix["grad_rows"] = {tok: (rows, g[tok][rows][:, :64].clone())}
            del g[tok]
        fix["grads"] = {k: (v[:64, :64] if v.dim() == 2 else v[:64]).clone() for k, v in g.items()}
    torch.save(fix, os.path.join(OUT, name + ".pt"))
    with torch.no_grad():
        h, p = O.text_tower(ids, mask, params, cfg)
    valid = mask.bool()
    print(f"{name}: stock-vs-oracle last(valid) {float((h - last)[valid].abs().max()):.2e} pooled "
          f"{float((p - pooled).abs().max()):.2e}; ref-encoder-vs-oracle pooled {float((p - pr).abs().max()):.2e}")


def fusion_fixture(name, modality_types, batch, feature_dims, fusion_dim, classes, seed, fusion_type="sum"):
    base = ref_shims.ref_baseline()
    args = types.SimpleNamespace(modality_types=list(modality_types), feature_dims=feature_dims, fusion_dim=fusion_dim,
                                 dropout_prob=0.0, fusion_type=fusion_type)

    class _Enc(torch.nn.Module):  # stands in for LanguageBind: returns the embeddings it is given
        def forward(self, data):
            return data

    model = base.finetune_model(args, classes, _Enc()).eval()
    head_in = fusion_dim if fusion_type in ("sum", "intra_attention", "inter_attention", "dedicated_dnn", "Distill_tea", "self_distill") else fusion_dim * len(modality_types)
    fp = O.init_fusion_params(modality_types, feature_dims, fusion_dim, classes, seed, head_in=head_in,
                              intra_attention=fusion_type == "intra_attention", dedicated=fusion_type == "dedicated_dnn",
                              regression=fusion_type == "regression", distillation=fusion_type in ("Distill_tea", "self_distill"),
                              inter_attention=fusion_type == "inter_attention")
    res = model.fusion.load_state_dict(fp, strict=False)
    assert not res.unexpected_keys and all(k.startswith("statistics_") for k in res.missing_keys), res        # (the concat head also carries statistics_<modal> buffers)
    g = _gen(seed + 1)
    stats = None
    if fusion_type == "concat":                            # test.py:112-115: mean / median embeddings of the training set
        stats = {m: torch.randn(feature_dims, generator=g) * 0.5 for m in modality_types}
        model.fusion.set_statistics({m: s.tolist() for m, s in stats.items()}, modality_types)
    emb = {m: torch.randn(batch, feature_dims, generator=g, requires_grad=True) for m in modality_types}
    codes = [0] + [base.missing_type_index[m] for m in modality_types]
    missing = torch.tensor([codes[i % len(codes)] for i in range(batch)], dtype=torch.int64)
    labels = torch.randint(0, classes, (batch,), generator=g)
    # (non-leaf copies, as encoder outputs are: modal_concat overwrites the missing rows of its inputs in place, :82)
    logits = model({m: e * 1.0 for m, e in emb.items()}, missing)
    features = cot_features = None
    extra = None
    if fusion_type == "self_distill":                      # training mode: (missing_mask, stu_features, tea_features, logits), :411
        model.train()
        masks, stu, tea, logits = model({m: e * 1.0 for m, e in emb.items()}, missing)
        cots = [torch.randn(t.shape, generator=g) * 0.01 for t in stu + [tea]]
        extra = {"masks": [m.clone() for m in masks], "stu": [t.detach() for t in stu], "tea": tea.detach(), "cots": cots}
        aux = sum((t * c).sum() for t, c in zip(stu + [tea], cots))
    elif isinstance(logits, tuple):                        # distillation heads return (features, logits), :380
        features, logits = logits
        cot_features = torch.randn(features.shape, generator=g) * 0.01
        aux = (features * cot_features).sum()
    else:
        aux = 0.0
    loss = torch.nn.CrossEntropyLoss()(logits, labels)
    (loss + aux).backward()
    fix = {"modality_types": list(modality_types), "params": fp, "emb": {m: e.detach() for m, e in emb.items()},
           "missing_index": missing, "labels": labels, "logits": logits.detach(), "loss": loss.detach(),
           "emb_grads": {m: e.grad.clone() for m, e in emb.items()},
           "grads": {k: v.grad.clone() for k, v in model.fusion.named_parameters()},
           "missing_type_index": dict(base.missing_type_index), "fusion_type": fusion_type, "statistics": stats,
           "features": None if features is None else features.detach(), "cot_features": cot_features, "self_distill": extra}
    torch.save(fix, os.path.join(OUT, name + ".pt"))
    with torch.no_grad():
        e0 = {m: e.detach() for m, e in emb.items()}
        if fusion_type == "sum":
            lo = O.fusion_sum(e0, missing, fp, modality_types)
        elif fusion_type == "intra_attention":
            lo = O.fusion_intra_attention(e0, missing, fp, modality_types)
        elif fusion_type == "inter_attention":
            lo = O.fusion_inter_attention(e0, missing, fp, modality_types)
        elif fusion_type == "dedicated_dnn":
            lo = O.fusion_dedicated_dnn(e0, missing, fp, modality_types)
        elif fusion_type == "regression":
            lo = O.fusion_regression(e0, missing, fp, modality_types)
        elif fusion_type == "Distill_tea":
            lo = O.fusion_distillation(e0, missing, fp, modality_types)[1]
        elif fusion_type == "self_distill":
            lo = O.fusion_self_distillation(e0, missing, fp, modality_types)[3]
        else:
            lo = O.fusion_concat(e0, missing, fp, modality_types, stats, mask=fusion_type == "concat")
    print(f"{name}: ref-vs-oracle logits {float((lo - logits).abs().max()):.2e}")


def bundle_fixture(name, seed):
    """LanguageBind.forward body (languagebind/__init__.py:75-85) run on a stand-in object that carries exactly
    the attributes that method touches; the method itself is the reference's, taken from its source file."""
    import ast
    src = open(ref_shims.REF_ROOT + "/languagebind/__init__.py").read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "LanguageBind")
    fwd = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "forward")
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fwd], type_ignores=[]), "languagebind/__init__.py", "exec"), ns)
    g = _gen(seed)
    d = 32
    pooled = {m: torch.randn(6, d, generator=g) for m in ("language", "image", "video")}
    proj = {m: torch.nn.Linear(d, d, bias=False) for m in pooled}
    for m in proj:
        proj[m].weight.data = torch.randn(d, d, generator=g) * d ** -0.5

    class _Tower(torch.nn.Module):
        def __init__(self, out):
            super().__init__()
            self.out = out

        def forward(self, **kw):
            return (None, self.out)

    self_ = types.SimpleNamespace(use_temp=True, modality_encoder={m: _Tower(pooled[m]) for m in pooled},
                                  modality_proj=proj,
                                  modality_scale={m: torch.tensor(2.6592) for m in pooled if m != "language"})
    with torch.no_grad():
        out = ns["forward"](self_, {m: {} for m in pooled})
    fix = {"pooled": pooled, "proj": {m: proj[m].weight.detach().clone() for m in proj}, "logit_scale": 2.6592,
           "out": {m: v.clone() for m, v in out.items()}}
    torch.save(fix, os.path.join(OUT, name + ".pt"))
    err = max(float((O.bundle_embed(pooled[m], fix["proj"][m], torch.tensor(2.6592), m) - out[m]).abs().max()) for m in out)
    print(f"{name}: ref-vs-oracle {err:.2e}")


def resize_pos_fixture(name, seed):
    """``resize_pos`` (image/modeling_image.py:795-839) run from the reference's source on a stand-in embeddings module: a 4 x 4
    position grid (+ class token) mapped onto the audio model's (num_mel_bins, target_length) patch grid."""
    import ast
    import math
    import torch.nn as nn
    import torch.nn.functional as F
    src = open(ref_shims.REF_ROOT + "/languagebind/image/modeling_image.py").read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "LanguageBindImage")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "resize_pos")
    ns = {"torch": torch, "nn": nn, "F": F, "math": math}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "modeling_image.py", "exec"), ns)
    g = _gen(seed)
    d, ps = 24, 16

    class _Emb(nn.Module):
        def __init__(self):
            super().__init__()
            self.image_size, self.patch_size, self.embed_dim = 64, ps, d
            self.config = types.SimpleNamespace(image_size=64)
            self.position_embedding = nn.Embedding(17, d)
            self.position_embedding.weight.data = torch.randn(17, d, generator=g)

    m = _Emb()
    old = m.position_embedding.weight.detach().clone()
    vc = types.SimpleNamespace(num_mel_bins=48, target_length=112)
    ns["resize_pos"](None, m, vc)
    new = m.position_embedding.weight.detach().clone()
    fix = {"old": old, "grid": (48 // ps, 112 // ps), "new": new, "num_mel_bins": 48, "target_length": 112, "patch_size": ps}
    torch.save(fix, os.path.join(OUT, name + ".pt"))
    print(f"{name}: ref-vs-oracle {float((O.resize_pos_embed(old, fix['grid']) - new).abs().max()):.2e}  shape {tuple(new.shape)}")


def losses_fixture(name, seed):
    """The distillation losses of the student training modes, from the reference's own source: class ``KL_loss``
    (train_ddp.py:70-79) is compiled from the file (the module itself needs tensorboard / datasets and is not importable), and
    the self-distillation loop body (train_ddp.py:235-242) is replayed with it on seeded tensors."""
    import ast
    import torch.nn as nn
    import torch.nn.functional as F
    src = open(ref_shims.REF_ROOT + "/train_ddp.py").read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "KL_loss")
    ns = {"torch": torch, "nn": nn, "F": F}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), "train_ddp.py", "exec"), ns)
    kl = ns["KL_loss"]()
    g = _gen(seed)
    B, C = 12, 32
    gs = (torch.randn(B, C, generator=g) * 0.3).requires_grad_(True)
    gt = torch.randn(B, C, generator=g) * 0.3
    l_kl = kl(gs, gt)
    l_kl.backward()
    fix = {"g_s": gs.detach().clone(), "g_t": gt, "temperature": kl.temperature, "kl": l_kl.detach(), "kl_grad": gs.grad.clone()}
    a = torch.randn(B, C, generator=g, requires_grad=True)
    b = torch.randn(B, C, generator=g)
    l_mse = nn.MSELoss()(a, b)
    l_mse.backward()
    fix.update(mse_a=a.detach().clone(), mse_b=b, mse=l_mse.detach(), mse_grad=a.grad.clone())
    # self-distillation loop body (train_ddp.py:235-242) on three "modalities"
    masks = [torch.rand(B, generator=g) > 0.3 for _ in range(3)]
    stu = [(torch.randn(B, C, generator=g) * 0.3).requires_grad_(True) for _ in range(3)]
    tea = (torch.randn(B, C, generator=g) * 0.3).requires_grad_(True)
    logits = torch.randn(B, 5, generator=g, requires_grad=True)
    labels = torch.randint(0, 5, (B,), generator=g)
    dl = 0
    for i, mask in enumerate(masks):
        t = tea[mask]
        s_ = stu[i][mask]
        dl += kl(s_, t)
    loss = 0.01 * dl / len(masks) + nn.CrossEntropyLoss()(logits, labels)
    loss.backward()
    fix["self_distill"] = {"masks": masks, "stu": [t.detach().clone() for t in stu], "tea": tea.detach().clone(), "logits": logits.detach().clone(),
                           "labels": labels, "loss": loss.detach(), "stu_grads": [t.grad.clone() for t in stu],
                           "tea_grad": None if tea.grad is None else tea.grad.clone(), "logits_grad": logits.grad.clone()}
    pt, ps = torch.randn(40, generator=g), torch.randn(40, generator=g)
    fix["ema"] = {"tea": pt, "stu": ps, "out": pt * 0.999 + ps * (1. - 0.999)}           # train_ddp.py:259
    torch.save(fix, os.path.join(OUT, name + ".pt"))
    e1 = float((O.kl_loss(fix["g_s"], gt, kl.temperature) - l_kl).abs())
    e2 = float((O.self_distill_loss(masks, fix["self_distill"]["stu"], fix["self_distill"]["tea"], fix["self_distill"]["logits"], labels) - loss).abs())
    print(f"{name}: ref-vs-oracle KL {e1:.2e} self-distill loss {e2:.2e}")


def missing_fixture(name):
    gm = ref_shims.ref_generate_missing()
    cases = []
    for n, ratio, modal, seed in [(32, 0.3, ["language", "video", "audio", "image"], 2025),
                                  (256, 0.3, ["language", "image"], 2026), (100, 0.7, ["video", "audio", "image"], 7)]:
        ref = gm.simulate_missing_modality(n, "mixed", ratio, modal + ["mixed"], seed)
        cases.append({"n": n, "ratio": ratio, "modal": modal, "seed": seed, "index": torch.tensor(ref, dtype=torch.int64)})
        mine = O.synth_missing_index(n, modal, ratio, seed)
        assert torch.equal(mine, cases[-1]["index"]), "oracle missing-index restatement differs"
    torch.save(cases, os.path.join(OUT, name + ".pt"))
    print(f"{name}: bit-exact on {len(cases)} cases")


def main():
    only = set(sys.argv[1:])        # optional: names of the fixtures to (re)generate; default all

    def run(fn, name, *a, **k):
        if not only or name in only:
            fn(name, *a, **k)

    tiny = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, image_size=32, patch_size=16)
    lname = "encoder.layers.0"
    vis_grads = ("embeddings.class_embedding", "embeddings.patch_embedding.weight", "embeddings.position_embedding.weight",
                 "pre_layrnorm.weight", "pre_layrnorm.bias", f"{lname}.self_attn.q_proj.weight", f"{lname}.self_attn.k_proj.bias",
                 f"{lname}.self_attn.v_proj.weight", f"{lname}.self_attn.out_proj.weight", f"{lname}.layer_norm1.weight",
                 f"{lname}.layer_norm2.bias", f"{lname}.mlp.fc1.weight", f"{lname}.mlp.fc1.bias", f"{lname}.mlp.fc2.weight",
                 "encoder.layers.1.mlp.fc2.bias", "post_layernorm.weight", "post_layernorm.bias")
    run(vision_fixture, "vision_tiny", "image", O.VisionCfg(**tiny), batch=3, seed_w=0, seed_x=1, grad_names=vis_grads)
    vid_grads = vis_grads + (f"{lname}.temporal_embedding", f"{lname}.temporal_attn.q_proj.weight",
                             f"{lname}.temporal_attn.out_proj.bias", f"{lname}.temporal_layer_norm1.weight")
    run(vision_fixture, "video_tiny", "video", O.VisionCfg(**tiny, add_time_attn=True, num_frames=4), batch=2, seed_w=3, seed_x=4,
                   grad_names=vid_grads)
    # the IMAGE-family time branch (image/modeling_image.py:83-84,129-134): the same layer with a temporal MLP behind the temporal
    # attention, which the video file removed - run on the reference's image tower with add_time_attn=True
    run(vision_fixture, "image_time_tiny", "image", O.VisionCfg(**tiny, add_time_attn=True, num_frames=4, temporal_mlp=True), batch=2,
        seed_w=13, seed_x=14, grad_names=vid_grads + (f"{lname}.temporal_mlp.fc1.weight", f"{lname}.temporal_mlp.fc2.bias",
                                                     f"{lname}.temporal_layer_norm2.weight", "encoder.layers.1.temporal_mlp.fc2.weight"))
    # PatchDropout in training mode (image/modeling_image.py:30-63): 64 x 64 images = 16 patch tokens, half of them dropped; once per
    # frame (T = 1) and once with the draw shared by a sample's four frames (T > 1)
    pd = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, image_size=64, patch_size=16)
    run(vision_fixture, "patch_dropout_tiny", "image", O.VisionCfg(**pd, force_patch_dropout=0.5), batch=3, seed_w=15, seed_x=16,
        grad_names=vis_grads)
    run(vision_fixture, "patch_dropout_video_tiny", "video", O.VisionCfg(**pd, add_time_attn=True, num_frames=4, force_patch_dropout=0.4),
        batch=2, seed_w=17, seed_x=18, grad_names=vid_grads)
    # a 197-token, head_dim-64 case small enough to commit: exercises the production attention shape
    run(vision_fixture, "vision_s197", "image",
                   O.VisionCfg(hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2,
                               image_size=224, patch_size=16), batch=2, seed_w=5, seed_x=6, store_params=False,
                   grad_names=(f"{lname}.self_attn.q_proj.weight", f"{lname}.mlp.fc1.bias"), compact=True)
    run(text_fixture, "text_tiny", O.TextCfg(vocab_size=512, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                        num_attention_heads=2, max_position_embeddings=16), batch=5, seed_w=7, seed_x=8,
                 grad_names=("embeddings.token_embedding.weight", "embeddings.position_embedding.weight",
                             f"{lname}.self_attn.q_proj.weight", f"{lname}.mlp.fc2.weight", "final_layer_norm.weight"))
    # BASELINE.json configs[1]'s text side at full size: d=768, 12 layers, 12 heads (head_dim 64), S=77, vocab 49408, causal + padding
    run(text_fixture, "text_full", O.TextCfg(), batch=4, seed_w=9, seed_x=10, compact=True,
        grad_names=("embeddings.token_embedding.weight", "embeddings.position_embedding.weight", f"{lname}.self_attn.q_proj.weight",
                    "encoder.layers.11.mlp.fc2.weight", "encoder.layers.5.self_attn.k_proj.weight", "final_layer_norm.weight"))
    run(fusion_fixture, "fusion_sum", ["language", "video", "audio", "image"], batch=10, feature_dims=48, fusion_dim=32, classes=5, seed=11)
    run(fusion_fixture, "fusion_concat", ["language", "video", "audio", "image"], batch=10, feature_dims=48, fusion_dim=32, classes=5, seed=21,
                   fusion_type="concat")
    run(fusion_fixture, "fusion_retrieval", ["language", "video", "image"], batch=9, feature_dims=48, fusion_dim=32, classes=4, seed=22,
                   fusion_type="retrieval")
    run(fusion_fixture, "fusion_intra_attention", ["language", "video", "audio", "image"], batch=10, feature_dims=48, fusion_dim=32, classes=5,
        seed=23, fusion_type="intra_attention")
    run(fusion_fixture, "fusion_inter_attention", ["language", "video", "audio", "image"], batch=10, feature_dims=48, fusion_dim=32, classes=5,
        seed=28, fusion_type="inter_attention")
    run(fusion_fixture, "fusion_dedicated_dnn", ["language", "video", "audio", "image"], batch=10, feature_dims=48, fusion_dim=32, classes=5,
        seed=24, fusion_type="dedicated_dnn")
    run(fusion_fixture, "fusion_regression", ["language", "video", "audio", "image"], batch=10, feature_dims=48, fusion_dim=32, classes=5,
        seed=25, fusion_type="regression")
    run(fusion_fixture, "fusion_distillation", ["language", "video", "audio", "image"], batch=10, feature_dims=48, fusion_dim=32, classes=5,
        seed=26, fusion_type="Distill_tea")
    run(fusion_fixture, "fusion_self_distill", ["language", "video", "image"], batch=9, feature_dims=48, fusion_dim=32, classes=4,
        seed=27, fusion_type="self_distill")
    run(bundle_fixture, "bundle", seed=12)
    run(losses_fixture, "distill_losses", seed=31)
    run(resize_pos_fixture, "resize_pos", seed=33)
    run(missing_fixture, "missing_index")
    # BASELINE.json configs[0]: image tower ViT-B/16 forward, B=4, 224x224 (weights by recipe, outputs stored)
    run(vision_fixture, "vitb16_config1", "image", O.VisionCfg(), batch=4, seed_w=0, seed_x=1, store_params=False, compact=True)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
