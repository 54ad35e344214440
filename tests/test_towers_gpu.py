"""Tower / bundle / fusion parity on the GPU: HIP path (through the drop-in modules and the C ABI) against
(a) the fixtures captured from the reference and (b) the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star): 1e-3 relative, fp32 instantiation.  The bf16 instantiation is checked at 3e-2."""
import types

import pytest
import torch

import missm_oracle as O
from conftest import load_golden

pytestmark = pytest.mark.gpu

TOL32 = 1e-3
TOLBF = 3e-2


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import missm_benchmark_amd as M
    lb, base = M.install()
    from missm_benchmark_amd import towers
    return types.SimpleNamespace(lb=lb, base=base, towers=towers)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.isfinite(a).all()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


# bf16 instantiation: relative Frobenius error ||g - ref|| / ||ref|| per class of tensor (VERDICT r2 #3b: cos / norm alone pass a
# gradient that drops 5 % of its rows or mis-scales one head of twelve - each of those is a >= 0.2 Frobenius error).  Measured over
# every gradient this suite checks (MISSM_GRAD_REPORT=<file> logs them; round 3, 186 bf16 tensors): median 5e-3 .. 7e-3, largest
# 2.9e-2 (full-size B = 32 step, the depth tower under 30 % missing codes); the fp32 instantiation of the same kernels: median 1e-6,
# largest 1.1e-5.  Bounds = 1.4x the largest measured value of the class:
#   matrix    GEMM weight gradients (q/k/v/out/fc1/fc2, patch embedding, projections, fusion linears): bf16 operands, fp32 sums
#   vector    bias / LayerNorm / class- and temporal-embedding gradients: column sums of bf16-rounded rows
#   embedding position / token embedding tables
#   time_qk   q_proj / k_proj WEIGHT gradients of the video tower's temporal attention: 8 keys with near-equal scores at init - the
#             loss barely depends on these two matrices (entries of ~1e-7), so what is left of their gradient is the residue of
#             cancelling terms and amplifies every rounding upstream: 0.17 .. 0.20 in bf16, and the fp32 instantiation's largest
#             error sits on the same tensor (1.1e-5 against a median of 1e-6: the same ~7x).  Direction (cos > 0.97) and norm
#             (+-10 %) are still held; the bound says "conditioning, not a dropped row": a dropped head would be > 0.28.
BF16_FRO = {"matrix": 4e-2, "vector": 4e-2, "embedding": 4e-2, "time_qk": 0.28}


def grad_class(name, t):
    if "position_embedding" in name or "token_embedding" in name:
        return "embedding"
    if "temporal_attn.q_proj.weight" in name or "temporal_attn.k_proj.weight" in name:
        return "time_qk"
    return "matrix" if t.dim() >= 2 and min(t.shape[0], t.shape[-1]) > 1 else "vector"


def fro(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def grad_ok(name, mine, ref, tol, dtype, fro_scale=1.0):
    """k_proj.bias has an identically-zero gradient (softmax is invariant to a per-query constant), so the reference
    value is rounding noise: compare it on an absolute scale."""
    import os
    if name.endswith("k_proj.bias"):
        return float((mine.detach().float().cpu() - ref.detach().float().cpu()).abs().max()) < (1e-5 if dtype == torch.float32 else 5e-3)
    rep = os.environ.get("MISSM_GRAD_REPORT")
    if rep:
        with open(rep, "a") as f:
            f.write(f"{'bf16' if dtype == torch.bfloat16 else 'f32 '} {grad_class(name, ref):9s} fro {fro(mine, ref):.3e} max {rel(mine, ref):.3e} "
                    f"{tuple(ref.shape)} {name} [{os.environ.get('PYTEST_CURRENT_TEST', '').split('::')[-1]}]\n")
    if dtype == torch.bfloat16:
        # bf16 operands: direction and magnitude (gradients that are small sums of cancelling terms - LayerNorm gains over 30 rows -
        # carry a few percent of rounding noise in their worst element) AND the Frobenius bound of the tensor's class
        a, b = mine.detach().float().cpu().flatten(), ref.detach().float().cpu().flatten()
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30))
        return (cos > 0.97 and abs(float(a.norm() / b.norm().clamp_min(1e-30)) - 1.0) < 0.1
                and fro(mine, ref) < BF16_FRO[grad_class(name, ref)] * fro_scale)
    return rel(mine, ref) < tol


def make_tower(pkg, cfgd, kind, params, dtype):
    fields = pkg.towers.TowerConfig.__dataclass_fields__
    cfg = pkg.towers.TowerConfig(kind=kind, **{k: v for k, v in cfgd.items() if k in fields and k != "kind"})
    t = pkg.towers.ClipTower(cfg, compute_dtype=dtype)
    res = t.load_state_dict(params, strict=True)
    return t.cuda()


def vision_inputs(fix, cfg):
    if "pixel_values" in fix:
        return fix["pixel_values"]
    shape = (fix["batch"], cfg.num_channels) + ((cfg.num_frames,) if cfg.num_frames > 1 else ()) + (cfg.image_size, cfg.image_size)
    return torch.randn(*shape, generator=torch.Generator().manual_seed(fix["seed_x"]))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL32), (torch.bfloat16, TOLBF)])
@pytest.mark.parametrize("name", ["vision_tiny", "video_tiny", "image_time_tiny", "patch_dropout_tiny", "patch_dropout_video_tiny", "vision_s197"])
def test_vision_tower_vs_reference_fixture(pkg, name, dtype, tol):
    fix = load_golden(name)
    ocfg = O.VisionCfg(**fix["cfg"])
    params = fix.get("params") or O.init_tower_params(ocfg, fix["seed_w"])
    tower = make_tower(pkg, fix["cfg"], "vision", params, dtype)
    x = vision_inputs(fix, ocfg)
    if "patch_keep" in fix:
        # PatchDropout (image/modeling_image.py:30-63): the reference ran in training mode; its kept-token indices are an input here.
        # In eval mode the layer is the identity, and the tower's own draw - the reference's torch.randn(...).topk(...) on the CPU
        # generator - reproduces the reference's choice under the reference's seed.
        assert tower.eval()(x.cuda())[0].shape[1] == ocfg.seq_len
        tower.train()
        torch.manual_seed(fix["patch_seed"])
        with torch.no_grad():
            own = tower(x.cuda())[1]
        last, pooled = tower(x.cuda(), patch_keep=fix["patch_keep"])
        assert torch.equal(own, pooled.detach()) and last.shape[1] == 1 + fix["patch_keep"].shape[1]
    else:
        last, pooled = tower(x.cuda())
    assert rel(pooled, fix["pooled"]) < tol
    if "last_hidden_state" in fix:
        assert rel(last, fix["last_hidden_state"]) < tol
        cp, ch = fix["cot_pooled"], fix["cot_last"]
    else:
        assert rel(last[:, :4, :64], fix["last_hidden_slice"]) < tol
        cp = torch.randn(pooled.shape, generator=torch.Generator().manual_seed(fix["seed_x"] + 100))
        ch = torch.randn(last.shape, generator=torch.Generator().manual_seed(fix["seed_x"] + 101)) * 0.1
    ((pooled * cp.cuda()).sum() + (last * ch.cuda()).sum()).backward()
    gtol = tol * (2 if dtype == torch.float32 else 3)
    for k, g in fix["grads"].items():
        mine = tower.get_parameter(k).grad
        assert mine is not None, k
        if mine.shape != g.shape:
            mine = mine[:64, :64] if mine.dim() == 2 else mine[:64]
        assert grad_ok(k, mine, g, gtol, dtype), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("name", ["vision_tiny", "video_tiny", "image_time_tiny", "patch_dropout_video_tiny", "vision_s197"])
def test_last_layer_backward_on_cls_rows_equals_dense(pkg, name, dtype):
    """When only the pooled output is differentiated (every training loop of the reference: `modality_encoder[key](**value)[1]`,
    languagebind/__init__.py:78) the last layer's MLP / out-projection backward runs on the CLS rows alone (towers.backward_lanes);
    a zero cotangent for last_hidden_state forces the all-rows path.  Same gradients: fp32 instantiation to 2e-5 of each tensor's scale
    (summation order), bf16 to 2e-2 (the dense path rounds the same products in another order)."""
    fix = load_golden(name)
    ocfg = O.VisionCfg(**fix["cfg"])
    params = fix.get("params") or O.init_tower_params(ocfg, fix["seed_w"])
    x = vision_inputs(fix, ocfg).cuda()
    kw = dict(patch_keep=fix["patch_keep"]) if "patch_keep" in fix else {}
    grads, pooleds = [], []
    for mode in ("cls_backward", "dense", "pooled_only_forward"):
        tower = make_tower(pkg, fix["cfg"], "vision", params, dtype)
        if kw:
            tower.train()
        if mode == "pooled_only_forward":
            # what the bundle does (languagebind/__init__.py:78 reads [1] only): the last layer's out-projection / MLP FORWARD on the CLS
            # rows as well, last_hidden_state left uncomputed
            with pkg.towers.pooled_output_only():
                last, pooled = tower(x, **kw)
            assert last is None
        else:
            last, pooled = tower(x, **kw)
        cp = torch.randn(pooled.shape, generator=torch.Generator().manual_seed(5)).cuda()
        loss = (pooled * cp).sum()
        if mode == "dense":
            loss = loss + (last * 0.0).sum()
        loss.backward()
        pooleds.append(pooled.detach().float().cpu())
        grads.append({k: p.grad.detach().float().cpu().clone() for k, p in tower.named_parameters() if p.grad is not None})
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert torch.equal(pooleds[0], pooleds[1]) and rel(pooleds[2], pooleds[1]) < tol
    for other in (0, 2):
        assert grads[other].keys() == grads[1].keys() and len(grads[1]) > 10
        for k in grads[1]:
            a, b = grads[other][k], grads[1][k]
            if k.endswith("k_proj.bias"):        # an identically-zero gradient (see grad_ok): rounding noise on every path
                assert float((a - b).abs().max()) < (1e-5 if dtype == torch.float32 else 5e-3), k
                continue
            assert float((a - b).abs().max()) <= tol * float(b.abs().max().clamp_min(1e-12)) + 1e-12, (other, k)


def test_full_size_vitb16_vs_reference_and_properties(pkg):
    """BASELINE.json configs[0] at full size on the GPU (ViT-B/16, 224x224, 197 tokens, B = 4): pooled output and a slice of
    the last hidden state against the fixture captured from the reference (fp32 instantiation 1e-3, bf16 3e-2), then the
    size-independent properties: batch independence, linearity of the backward in the upstream gradient, and run-to-run
    reproducibility of the weight gradients (split-K slices are summed in a fixed order)."""
    fix = load_golden("vitb16_config1")
    ocfg = O.VisionCfg(**fix["cfg"])
    params = O.init_tower_params(ocfg, fix["seed_w"])
    x = vision_inputs(fix, ocfg).cuda()
    t32 = make_tower(pkg, fix["cfg"], "vision", params, torch.float32)
    with torch.no_grad():
        last32, pooled32 = t32(x)
    assert rel(pooled32, fix["pooled"]) < TOL32 and rel(last32[:, :4, :64], fix["last_hidden_slice"]) < TOL32
    del t32
    tower = make_tower(pkg, fix["cfg"], "vision", params, torch.bfloat16)
    last, pooled = tower(x)
    assert rel(pooled, fix["pooled"]) < TOLBF and rel(last[:, :4, :64], fix["last_hidden_slice"]) < TOLBF
    with torch.no_grad():                                   # batch independence: samples do not see each other
        _, half = tower(x[:2])
    assert rel(half, pooled[:2]) < 2e-3
    cot = torch.randn(pooled.shape, generator=torch.Generator().manual_seed(5)).cuda()
    names = ["encoder.layers.0.self_attn.q_proj.weight", "encoder.layers.11.mlp.fc2.weight", "embeddings.patch_embedding.weight",
             "encoder.layers.5.mlp.fc1.bias", "post_layernorm.weight"]
    (pooled * cot).sum().backward()
    g1 = {k: tower.get_parameter(k).grad.clone() for k in names}
    _, pooled_b = tower(x)
    assert torch.equal(pooled_b, pooled)                    # forward is deterministic
    tower.zero_grad(set_to_none=True)                       # (without it a second backward accumulates, like autograd)
    (pooled_b * (2.0 * cot)).sum().backward()               # gradients are linear in the cotangent
    for k in names:
        g2 = tower.get_parameter(k).grad
        assert rel(g2, 2.0 * g1[k]) < 1e-3, k
    _, pooled_c = tower(x)
    tower.zero_grad(set_to_none=True)
    (pooled_c * (2.0 * cot)).sum().backward()
    g3 = {k: tower.get_parameter(k).grad.clone() for k in names[:3]}   # weight gradients (GEMM + ordered split-K reduce):
    _, pooled_d = tower(x)
    tower.zero_grad(set_to_none=True)
    (pooled_d * (2.0 * cot)).sum().backward()
    for k in names[:3]:                                     # ... bit-reproducible from run to run
        assert torch.equal(tower.get_parameter(k).grad, g3[k]), k


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL32), (torch.bfloat16, TOLBF)])
def test_text_tower_vs_fixture(pkg, dtype, tol):
    fix = load_golden("text_tiny")
    tower = make_tower(pkg, fix["cfg"], "text", fix["params"], dtype)
    ids, mask = fix["input_ids"].cuda(), fix["attention_mask"].cuda()
    last, pooled = tower(input_ids=ids, attention_mask=mask)
    assert rel(pooled, fix["pooled"]) < tol
    valid = fix["attention_mask"].bool()
    assert rel(last.cpu()[valid], fix["last_hidden_state"][valid]) < tol
    (pooled * fix["cot_pooled"].cuda()).sum().backward()
    for k, g in fix["grads"].items():
        assert grad_ok(k, tower.get_parameter(k).grad, g, tol * 3, dtype), k
    with pytest.raises(ValueError, match="You have to specify input_ids"):
        tower()


def test_vision_errors_and_input_ranks(pkg):
    fix = load_golden("video_tiny")
    tower = make_tower(pkg, fix["cfg"], "vision", fix["params"], torch.float32)
    with pytest.raises(ValueError, match="You have to specify pixel_values"):
        tower(None)
    x5 = fix["pixel_values"]                     # [B, C, T, H, W]
    B, C, T, H, W = x5.shape
    with torch.no_grad():
        _, p5 = tower(x5.cuda())
        x7 = x5.permute(0, 2, 1, 3, 4).reshape(B, 1, T, 1, C, H, W)   # (b, pair, T, bs, C, H, W) flattens to (b t) frames
        _, p7 = tower(x7.cuda())
    assert rel(p7, p5) < 1e-6
    with pytest.raises(ValueError):
        tower(x5[:, :, :2].cuda())               # wrong frame count for the time attention


def test_state_dict_roundtrip_and_device_move(pkg):
    fix = load_golden("vision_tiny")
    tower = make_tower(pkg, fix["cfg"], "vision", fix["params"], torch.float32)
    sd = tower.state_dict()
    assert set(sd) == set(fix["params"])
    for k, v in fix["params"].items():
        assert torch.equal(sd[k].cpu(), v), k
    # parameters stay views of one flat allocation after .cuda()
    base = tower.flat_master()
    assert all(p.data_ptr() >= base.data_ptr() and p.data_ptr() < base.data_ptr() + base.numel() * 4 for p in tower.parameters())
    # an in-place parameter update (what an optimizer does) is picked up by the compute-dtype shadows
    x = fix["pixel_values"].cuda()
    with torch.no_grad():
        _, p0 = tower(x)
        tower.get_parameter("encoder.layers.0.mlp.fc1.weight").mul_(1.5)
        _, p1 = tower(x)
    assert rel(p1, p0) > 1e-4


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL32)])
def test_fusion_sum_and_bundle_vs_reference_fixture(pkg, dtype, tol):
    fix = load_golden("fusion_sum")
    mt = fix["modality_types"]
    fd = fix["params"]["modal_proj." + mt[0] + ".weight"].shape
    args = types.SimpleNamespace(modality_types=mt, feature_dims=fd[1], fusion_dim=fd[0], dropout_prob=0.0, fusion_type="sum")
    C = fix["logits"].shape[1]
    fusion = pkg.base.modal_sum(args, C)
    fusion.load_state_dict(fix["params"], strict=True)
    fusion = fusion.cuda()
    emb = {m: e.cuda().requires_grad_(True) for m, e in fix["emb"].items()}
    logits = fusion(emb, fix["missing_index"].cuda())
    assert rel(logits, fix["logits"]) < 1e-5
    from missm_benchmark_amd.nn import HipCrossEntropyLoss
    loss = HipCrossEntropyLoss()(logits, fix["labels"].cuda())
    assert abs(float(loss) - float(fix["loss"])) < 1e-5
    loss.backward()
    for m in mt:
        assert rel(emb[m].grad, fix["emb_grads"][m]) < 1e-4, m
    for k, g in fix["grads"].items():
        assert rel(fusion.get_parameter(k).grad, g) < 1e-4, k
    assert pkg.base.missing_type_index["language"] == 1 and pkg.base.missing_type_index["image"] == 4
    # bundle: projection -> L2 normalise -> temperature (languagebind/__init__.py:78-84)
    bf = load_golden("bundle")
    from missm_benchmark_amd import nn as hnn
    for m, ref in bf["out"].items():
        w = bf["proj"][m].cuda()
        e = hnn._LinearFn.apply(bf["pooled"][m].cuda(), w, None, False, None, 0)
        scale = 1.0 if m == "language" else float(torch.tensor(bf["logit_scale"]).exp())
        assert rel(hnn.l2norm_scale(e, scale), ref) < 1e-5


@pytest.mark.parametrize("name", ["fusion_concat", "fusion_retrieval", "fusion_intra_attention", "fusion_inter_attention",
                                  "fusion_dedicated_dnn", "fusion_regression", "fusion_distillation"])
def test_fusion_concat_heads_vs_reference_fixture(pkg, name):
    """fusion_type 'concat' (zero / mean / median imputation through set_statistics, test.py:112-115) and 'retrieval':
    projections written straight into their slice of the concatenated row; logits, loss and every gradient against the
    fixture captured from the reference."""
    fix = load_golden(name)
    mt = fix["modality_types"]
    fdim, cdim = fix["params"]["head.head.3.weight"].shape[1], next(iter(fix["emb"].values())).shape[1]
    args = types.SimpleNamespace(modality_types=mt, feature_dims=cdim, fusion_dim=fdim, dropout_prob=0.0, fusion_type=fix["fusion_type"])
    C = fix["logits"].shape[1]
    model = pkg.base.finetune_model(args, C, torch.nn.Identity())
    assert type(model.fusion).__name__ == {"concat": "modal_concat", "retrieval": "modal_concat_full",
                                           "intra_attention": "modal_intra_channel_attention", "inter_attention": "modal_inter_attention",
                                           "dedicated_dnn": "modal_dedicated_dnn", "regression": "modal_regression",
                                           "Distill_tea": "modal_distillation"}[fix["fusion_type"]]
    missing, unexpected = model.fusion.load_state_dict(fix["params"], strict=False)
    assert not unexpected and all(k.startswith("statistics_") for k in missing)
    model = model.cuda()
    if fix["statistics"] is not None:
        model.fusion.set_statistics({m: s.tolist() for m, s in fix["statistics"].items()}, mt)
        assert all(f"statistics_{m}" in model.fusion.state_dict() for m in mt)
    emb = {m: e.cuda().requires_grad_(True) for m, e in fix["emb"].items()}
    logits = model(emb, fix["missing_index"].cuda())
    feats = None
    if isinstance(logits, tuple):                     # distillation heads: (features, logits)
        feats, logits = logits
        assert rel(feats, fix["features"]) < 1e-6
    assert rel(logits, fix["logits"]) < 1e-5
    from missm_benchmark_amd.nn import HipCrossEntropyLoss
    loss = HipCrossEntropyLoss()(logits, fix["labels"].cuda())
    assert abs(float(loss.detach()) - float(fix["loss"])) < 1e-5
    if feats is not None:
        feats.backward(fix["cot_features"].cuda(), retain_graph=True)     # the distillation losses' path into the features
    loss.backward()
    for m in mt:
        assert rel(emb[m].grad, fix["emb_grads"][m]) < 1e-4, m
    for k, g in fix["grads"].items():
        assert rel(model.fusion.get_parameter(k).grad, g) < 1e-4, k


def _tiny_model(pkg, dtype, seed=0):
    T = pkg.towers.TowerConfig
    tiny = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, image_size=32, patch_size=16)
    cfgs = {"video": T(kind="vision", add_time_attn=True, num_frames=4, **tiny), "image": T(kind="vision", **tiny)}
    tcfg = T(kind="text", hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, vocab_size=512,
             max_position_embeddings=16)
    enc = pkg.lb.LanguageBind({"video": "LanguageBind_Video", "image": "LanguageBind_Image"}, configs=cfgs, text_config=tcfg,
                              projection_dim=48, compute_dtype=dtype, seed=seed)
    args = types.SimpleNamespace(modality_types=["language", "video", "image"], feature_dims=48, fusion_dim=32, dropout_prob=0.0,
                                 fusion_type="sum")
    return pkg.base.finetune_model(args, 5, enc), cfgs, tcfg, args


def _oracle_of(model, cfgs, tcfg, args):
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    tp, tc = {}, {}
    for m in ("video", "image", "language"):
        pre = f"encoder.modality_encoder.{m}."
        tp[m] = {k[len(pre):]: v.requires_grad_(True) for k, v in sd.items() if k.startswith(pre)}
    oc = lambda c: O.VisionCfg(**{k: getattr(c, k) for k in O.VisionCfg.__dataclass_fields__})
    tc = {"video": oc(cfgs["video"]), "image": oc(cfgs["image"]),
          "language": O.TextCfg(**{k: getattr(tcfg, k) for k in O.TextCfg.__dataclass_fields__})}
    proj = {m: sd[f"encoder.modality_proj.{m}.weight"].requires_grad_(True) for m in tp}
    scales = {m: torch.tensor(2.6592) for m in ("video", "image")}
    fp = {k[len("fusion."):]: v.requires_grad_(True) for k, v in sd.items() if k.startswith("fusion.")}
    return tp, tc, proj, scales, fp


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL32), (torch.bfloat16, 5e-2)])
def test_finetune_model_end_to_end_vs_oracle(pkg, dtype, tol):
    """finetune_model.forward + CE loss + full backward on 3 modalities (text, video with time attention, image)
    with one missing-modality code per sample, against the CPU oracle."""
    model, cfgs, tcfg, args = _tiny_model(pkg, dtype)
    tp, tc, proj, scales, fp = _oracle_of(model, cfgs, tcfg, args)
    model = model.cuda()
    B = 6
    g = torch.Generator().manual_seed(5)
    ids, mask = O.synth_text_batch(B, 16, 3, vocab=512)
    data = {"language": {"input_ids": ids, "attention_mask": mask},
            "video": {"pixel_values": torch.randn(B, 3, 4, 32, 32, generator=g)},
            "image": {"pixel_values": torch.randn(B, 3, 32, 32, generator=g)}}
    missing = torch.tensor([0, 1, 2, 4, 0, 2])
    labels = torch.randint(0, 5, (B,), generator=g)
    ologits, oemb = O.finetune_forward(data, missing, tp, tc, proj, scales, fp, args.modality_types)
    oloss = O.cross_entropy(ologits, labels)
    oloss.backward()
    from missm_benchmark_amd.nn import HipCrossEntropyLoss
    gdata = {m: {k: v.cuda() for k, v in d.items()} for m, d in data.items()}
    logits = model(gdata, missing.cuda())
    loss = HipCrossEntropyLoss()(logits, labels.cuda())
    loss.backward()
    assert rel(logits, ologits) < tol
    assert abs(float(loss) - float(oloss)) < tol * max(1.0, float(oloss))
    checks = {"encoder.modality_encoder.video.encoder.layers.0.temporal_attn.q_proj.weight": tp["video"]["encoder.layers.0.temporal_attn.q_proj.weight"],
              "encoder.modality_encoder.video.encoder.layers.1.mlp.fc1.weight": tp["video"]["encoder.layers.1.mlp.fc1.weight"],
              "encoder.modality_encoder.video.embeddings.patch_embedding.weight": tp["video"]["embeddings.patch_embedding.weight"],
              "encoder.modality_encoder.image.encoder.layers.0.self_attn.k_proj.bias": tp["image"]["encoder.layers.0.self_attn.k_proj.bias"],
              "encoder.modality_encoder.image.pre_layrnorm.weight": tp["image"]["pre_layrnorm.weight"],
              "encoder.modality_encoder.language.embeddings.token_embedding.weight": tp["language"]["embeddings.token_embedding.weight"],
              "encoder.modality_encoder.language.encoder.layers.0.layer_norm1.weight": tp["language"]["encoder.layers.0.layer_norm1.weight"],
              "encoder.modality_proj.video.weight": proj["video"], "fusion.modal_proj.image.weight": fp["modal_proj.image.weight"],
              "fusion.head.head.3.bias": fp["head.head.3.bias"]}
    gt = tol * (3 if dtype == torch.float32 else 4)
    for k, ref in checks.items():
        assert grad_ok(k, model.get_parameter(k).grad, ref.grad, gt, dtype), k


def test_full_size_five_modality_step_vs_oracle(pkg):
    """BASELINE.json configs[2]/[3] at full tower size (five ViT-B/16 towers, video with T = 8 factorised time attention,
    sum fusion under missing-modality codes, CE loss, full backward) at B = 2 against the CPU oracle: fp32 instantiation within
    1e-3 (north_star), bf16 instantiation by direction / magnitude."""
    mods = ["video", "image", "audio", "depth", "thermal"]
    T = pkg.towers.TowerConfig
    cfgs = {m: T(kind="vision", add_time_attn=(m == "video"), num_frames=8 if m == "video" else 1) for m in mods}
    enc = pkg.lb.LanguageBind({m: f"LanguageBind_{m.capitalize()}" for m in mods}, configs=cfgs, compute_dtype=torch.float32, seed=3)
    args = types.SimpleNamespace(modality_types=mods, feature_dims=768, fusion_dim=256, dropout_prob=0.0, fusion_type="sum")
    torch.manual_seed(0)
    model = pkg.base.finetune_model(args, 8, enc)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ocfg = {m: O.VisionCfg(add_time_attn=(m == "video"), num_frames=8 if m == "video" else 1) for m in mods}
    tp = {m: {k[len(f"encoder.modality_encoder.{m}."):]: v.requires_grad_(True) for k, v in sd.items()
              if k.startswith(f"encoder.modality_encoder.{m}.")} for m in mods}
    proj = {m: sd[f"encoder.modality_proj.{m}.weight"].requires_grad_(True) for m in mods}
    scales = {m: torch.tensor(2.6592) for m in mods}
    fp = {k[len("fusion."):]: v.requires_grad_(True) for k, v in sd.items() if k.startswith("fusion.")}
    B = 2
    g = torch.Generator().manual_seed(9)
    data = {m: {"pixel_values": torch.randn(*((B, 3, 8, 224, 224) if m == "video" else (B, 3, 224, 224)), generator=g)} for m in mods}
    missing = torch.tensor([0, pkg.base.missing_type_index["audio"]])
    labels = torch.tensor([3, 5])
    torch.set_num_threads(max(1, min(16, len(__import__("os").sched_getaffinity(0)))))
    ologits, _ = O.finetune_forward(data, missing, tp, ocfg, proj, scales, fp, mods)
    oloss = O.cross_entropy(ologits, labels)
    oloss.backward()
    names = ["encoder.modality_encoder.video.encoder.layers.0.temporal_attn.q_proj.weight",
             "encoder.modality_encoder.video.encoder.layers.11.mlp.fc2.weight",
             "encoder.modality_encoder.video.embeddings.patch_embedding.weight",
             "encoder.modality_encoder.image.encoder.layers.5.self_attn.out_proj.weight",
             "encoder.modality_encoder.thermal.encoder.layers.0.mlp.fc1.bias",
             "encoder.modality_encoder.depth.post_layernorm.weight",
             "encoder.modality_proj.video.weight", "fusion.modal_proj.image.weight", "fusion.head.head.3.bias"]

    def oracle_grad(k):
        if k.startswith("encoder.modality_encoder."):
            m = k.split(".")[2]
            return tp[m][k[len(f"encoder.modality_encoder.{m}."):]].grad
        if k.startswith("encoder.modality_proj."):
            return proj[k.split(".")[2]].grad
        return fp[k[len("fusion."):]].grad

    from missm_benchmark_amd.nn import HipCrossEntropyLoss
    model = model.cuda()
    gdata = {m: {k: v.cuda() for k, v in d.items()} for m, d in data.items()}
    for dtype, tol in ((torch.float32, TOL32), (torch.bfloat16, 5e-2)):
        enc.set_compute_dtype(dtype)
        for p in model.parameters():
            p.grad = None
        logits = model(gdata, missing.cuda())
        loss = HipCrossEntropyLoss()(logits, labels.cuda())
        loss.backward()
        assert rel(logits, ologits) < tol, dtype
        assert abs(float(loss.detach()) - float(oloss.detach())) < tol * max(1.0, float(oloss.detach()))
        for k in names:
            assert grad_ok(k, model.get_parameter(k).grad, oracle_grad(k), tol * 4, dtype), (k, dtype)
    # the audio tower of sample 1 is marked missing: its embedding gets no gradient from that sample, and the audio
    # projection's weight gradient therefore equals the one-sample gradient (checked through the oracle above)


def test_self_distillation_head_vs_reference_fixture(pkg):
    """fusion_type 'self_distill': training-mode 4-tuple (missing masks, student features per modality, teacher features,
    logits) and all gradients against the fixture captured from the reference; eval mode returns the logits alone."""
    fix = load_golden("fusion_self_distill")
    mt = fix["modality_types"]
    fdim, cdim = fix["params"]["head.head.3.weight"].shape[1], next(iter(fix["emb"].values())).shape[1]
    args = types.SimpleNamespace(modality_types=mt, feature_dims=cdim, fusion_dim=fdim, dropout_prob=0.0, fusion_type="self_distill")
    model = pkg.base.finetune_model(args, fix["logits"].shape[1], torch.nn.Identity())
    model.fusion.load_state_dict(fix["params"], strict=True)
    model = model.cuda().train()
    emb = {m: e.cuda().requires_grad_(True) for m, e in fix["emb"].items()}
    masks, stu, tea, logits = model(emb, fix["missing_index"].cuda())
    sd = fix["self_distill"]
    assert all(torch.equal(a.cpu(), b) for a, b in zip(masks, sd["masks"]))
    assert all(rel(a, b) < 1e-5 for a, b in zip(stu, sd["stu"])) and rel(tea, sd["tea"]) < 1e-5 and rel(logits, fix["logits"]) < 1e-5
    from missm_benchmark_amd.nn import HipCrossEntropyLoss
    loss = HipCrossEntropyLoss()(logits, fix["labels"].cuda())
    torch.autograd.backward([loss] + stu + [tea], [None] + [c.cuda() for c in sd["cots"]])
    for m in mt:
        assert rel(emb[m].grad, fix["emb_grads"][m]) < 1e-4, m
    for k, g in fix["grads"].items():
        assert rel(model.fusion.get_parameter(k).grad, g) < 1e-4, k
    model.eval()
    with torch.no_grad():
        assert rel(model({m: e.detach() for m, e in emb.items()}, fix["missing_index"].cuda()), fix["logits"]) < 1e-5


@pytest.mark.parametrize("ftype", ["graph_fusion", "unified_graph"])
def test_graph_fusion_heads_vs_oracle(pkg, ftype):
    """fusion_type 'graph_fusion' / 'unified_graph' (reference src/model/baseline.py:240-331) on the SuperGAT kernels against the CPU
    oracle's dense restatement - logits, loss, every gradient.  PARITY UNPINNED: torch_geometric is absent and unpinned upstream, the
    oracle restates SuperGATConv from the published algorithm; no reference fixture exists for these two heads."""
    mt = ["language", "video", "audio", "image"]
    args = types.SimpleNamespace(modality_types=mt, feature_dims=768, fusion_dim=256, dropout_prob=0.0, fusion_type=ftype)
    torch.manual_seed(5)
    model = pkg.base.finetune_model(args, 5, torch.nn.Identity())
    with torch.no_grad():                        # (biases start at zero in torch_geometric: make them matter)
        for n, p in model.fusion.named_parameters():
            if n.endswith(".bias") and "gcn" in n:
                p.normal_(0.0, 0.05)
    fp = {k: v.detach().clone().requires_grad_(True) for k, v in model.fusion.state_dict().items()}
    model = model.cuda()
    B = 10
    g = torch.Generator().manual_seed(6)
    emb_cpu = {m: torch.randn(B, 768, generator=g) for m in mt}
    codes = [0] + [pkg.base.missing_type_index[m] for m in mt]
    missing = torch.tensor([codes[i % len(codes)] for i in range(B)], dtype=torch.int64)
    labels = torch.randint(0, 5, (B,), generator=g)
    oemb = {m: e.clone().requires_grad_(True) for m, e in emb_cpu.items()}
    ologits = (O.fusion_graph if ftype == "graph_fusion" else O.fusion_unified_graph)(oemb, missing, fp, mt)
    oloss = O.cross_entropy(ologits, labels)
    oloss.backward()
    emb = {m: e.cuda().requires_grad_(True) for m, e in emb_cpu.items()}
    logits = model(emb, missing.cuda())
    assert rel(logits, ologits) < 1e-4
    from missm_benchmark_amd.nn import HipCrossEntropyLoss
    loss = HipCrossEntropyLoss()(logits, labels.cuda())
    assert abs(float(loss.detach()) - float(oloss.detach())) < 1e-4
    loss.backward()
    for m in mt:
        assert rel(emb[m].grad, oemb[m].grad) < 1e-3, m
    for k, ref in fp.items():
        assert ref.grad is not None, k
        assert rel(model.fusion.get_parameter(k).grad, ref.grad) < 1e-3, k
