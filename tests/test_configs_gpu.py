"""BASELINE.json configs[1] and configs[3] at full tower size on the GPU, through the drop-in modules and the C ABI.

  * the full-size text tower (d=768, 12 layers, head_dim 64 -> the MFMA causal + key-mask attention kernel, S=77, vocab 49408,
    atomic scatter into the 49408 x 768 embedding gradient, EOT-gather LayerNorm) against the fixture captured from stock
    transformers + the reference's encoder code (tests/golden/text_full.pt) - fp32 instantiation 1e-3, bf16 by its own bars;
  * configs[1]: image + text `sum` fusion step against the CPU oracle at B = 4; at B = 32 through size-independent properties;
  * configs[3]: five modalities with 30 % mixed missing codes (incl. this build's codes 5 / 6) at B = 32: a missing
    (sample, modality) contributes exactly zero to that tower's embedding gradient; oracle agreement on 4 of the samples;
  * semantics the reference relies on: weights changed through torch (load_state_dict, torch.optim) reach the kernels,
    `.grad` accumulates over backward calls like autograd's.
"""
import os
import types

import pytest
import torch

import missm_oracle as O
from conftest import load_golden
from test_towers_gpu import TOL32, TOLBF, grad_ok, make_tower, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import missm_benchmark_amd as M
    lb, base = M.install()
    from missm_benchmark_amd import towers
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    return types.SimpleNamespace(lb=lb, base=base, towers=towers)


def _sliced(mine, ref):
    if mine.shape != ref.shape:
        mine = mine[:64, :64] if mine.dim() == 2 else mine[:64]
    return mine


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL32), (torch.bfloat16, TOLBF)])
def test_full_size_text_tower_vs_reference_fixture(pkg, dtype, tol):
    fix = load_golden("text_full")
    cfg = O.TextCfg(**fix["cfg"])
    params = O.init_tower_params(cfg, fix["seed_w"], kind="text")
    tower = make_tower(pkg, fix["cfg"], "text", params, dtype)
    assert tower.config.hidden_size // tower.config.num_attention_heads == 64      # the MFMA attention path
    ids, mask = fix["input_ids"].cuda(), fix["attention_mask"].cuda()
    last, pooled = tower(input_ids=ids, attention_mask=mask)
    assert rel(pooled, fix["pooled"]) < tol
    assert rel(pooled, fix["pooled_ref_encoder"]) < tol
    valid = fix["attention_mask"].bool()
    assert rel(last.cpu()[:, :, :64][valid], fix["last_hidden_slice"][valid]) < tol
    (pooled * fix["cot_pooled"].cuda()).sum().backward()
    gtol = tol * 3
    for k, g in fix["grads"].items():
        assert grad_ok(k, _sliced(tower.get_parameter(k).grad, g), g, gtol, dtype), k
    for k, (rows, g) in fix["grad_rows"].items():      # the embedding rows that received a scattered gradient
        mine = tower.get_parameter(k).grad[rows.cuda()][:, :64]
        assert grad_ok(k, mine, g, gtol, dtype), k
    # rows of the 49408-row table that no token of the batch addresses stay exactly zero
    used = torch.zeros(cfg.vocab_size, dtype=torch.bool)
    used[fix["input_ids"].flatten()] = True
    tg = tower.get_parameter("embeddings.token_embedding.weight").grad
    assert float(tg[(~used).cuda()].abs().max()) == 0.0


def _two_modality_model(pkg, dtype, seed=3):
    T = pkg.towers.TowerConfig
    enc = pkg.lb.LanguageBind({"image": "LanguageBind_Image"}, configs={"image": T(kind="vision")}, text_config=T(kind="text"),
                              compute_dtype=dtype, seed=seed)
    args = types.SimpleNamespace(modality_types=["language", "image"], feature_dims=768, fusion_dim=256, dropout_prob=0.0,
                                 fusion_type="sum")
    torch.manual_seed(0)
    return pkg.base.finetune_model(args, 8, enc), args


def _oracle_parts(sd, mods):
    tp, cfgs = {}, {}
    for m in mods:
        pre = f"encoder.modality_encoder.{m}."
        tp[m] = {k[len(pre):]: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith(pre)}
        cfgs[m] = O.TextCfg() if m == "language" else O.VisionCfg(add_time_attn=(m == "video"), num_frames=8 if m == "video" else 1)
    proj = {m: sd[f"encoder.modality_proj.{m}.weight"].clone().requires_grad_(True) for m in mods}
    scales = {m: torch.tensor(2.6592) for m in mods if m != "language"}
    fp = {k[len("fusion."):]: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("fusion.")}
    return tp, cfgs, proj, scales, fp


def _config1_batch(B, seed):
    g = torch.Generator().manual_seed(seed)
    ids, mask = O.synth_text_batch(B, 77, seed + 1)
    data = {"language": {"input_ids": ids, "attention_mask": mask},
            "image": {"pixel_values": torch.randn(B, 3, 224, 224, generator=g)}}
    return data, torch.randint(0, 8, (B,), generator=g)


def _to_gpu(data):
    return {m: {k: v.cuda() for k, v in d.items()} for m, d in data.items()}


def test_config1_image_text_step_vs_oracle(pkg):
    """configs[1] (image + text, `sum` fusion, C = 8) - forward, CE, full backward at B = 4 against the CPU oracle, both
    instantiations; one missing-language and one missing-image sample."""
    from missm_benchmark_amd.nn import HipCrossEntropyLoss
    model, args = _two_modality_model(pkg, torch.float32)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    tp, cfgs, proj, scales, fp = _oracle_parts(sd, ["language", "image"])
    data, labels = _config1_batch(4, 21)
    missing = torch.tensor([0, 1, 4, 0])
    ologits, _ = O.finetune_forward(data, missing, tp, cfgs, proj, scales, fp, args.modality_types)
    oloss = O.cross_entropy(ologits, labels)
    oloss.backward()
    og = {"encoder.modality_encoder.language.embeddings.token_embedding.weight": tp["language"]["embeddings.token_embedding.weight"].grad,
          "encoder.modality_encoder.language.encoder.layers.0.self_attn.q_proj.weight": tp["language"]["encoder.layers.0.self_attn.q_proj.weight"].grad,
          "encoder.modality_encoder.language.encoder.layers.11.mlp.fc1.weight": tp["language"]["encoder.layers.11.mlp.fc1.weight"].grad,
          "encoder.modality_encoder.language.final_layer_norm.weight": tp["language"]["final_layer_norm.weight"].grad,
          "encoder.modality_encoder.image.encoder.layers.6.self_attn.out_proj.weight": tp["image"]["encoder.layers.6.self_attn.out_proj.weight"].grad,
          "encoder.modality_encoder.image.embeddings.patch_embedding.weight": tp["image"]["embeddings.patch_embedding.weight"].grad,
          "encoder.modality_proj.language.weight": proj["language"].grad, "encoder.modality_proj.image.weight": proj["image"].grad,
          "fusion.modal_proj.language.weight": fp["modal_proj.language.weight"].grad, "fusion.head.head.3.bias": fp["head.head.3.bias"].grad}
    model = model.cuda()
    gdata = _to_gpu(data)
    for dtype, tol in ((torch.float32, TOL32), (torch.bfloat16, 5e-2)):
        model.encoder.set_compute_dtype(dtype)
        model.zero_grad(set_to_none=True)
        logits = model(gdata, missing.cuda())
        loss = HipCrossEntropyLoss()(logits, labels.cuda())
        loss.backward()
        assert rel(logits, ologits) < tol, dtype
        assert abs(float(loss.detach()) - float(oloss.detach())) < tol * max(1.0, float(oloss.detach()))
        for k, ref in og.items():
            mine = model.get_parameter(k).grad
            if "token_embedding" in k:          # compare the rows that carry gradient (a dense max over 38 M zeros says nothing)
                rows = torch.unique(data["language"]["input_ids"])
                mine, ref_ = mine[rows.cuda()], ref[rows]
            else:
                ref_ = ref
            assert grad_ok(k, mine, ref_, tol * 4, dtype), (k, dtype)


def test_config1_b32_properties(pkg):
    """configs[1] at its full batch (B = 32, bf16): batch independence against the B = 4 run the oracle checked, linearity of the
    backward in the loss scale, run-to-run bit-reproducibility of the GEMM weight gradients, zero gradient into untouched
    embedding rows."""
    from missm_benchmark_amd.nn import HipCrossEntropyLoss
    model, args = _two_modality_model(pkg, torch.bfloat16)
    model = model.cuda()
    data, labels = _config1_batch(32, 40)
    gdata = _to_gpu(data)
    missing = torch.zeros(32, dtype=torch.int64)
    missing[3], missing[17] = 1, 4
    crit = HipCrossEntropyLoss()
    logits = model(gdata, missing.cuda())
    with torch.no_grad():
        sub = {m: {k: v[:4] for k, v in d.items()} for m, d in gdata.items()}
        l4 = model(sub, missing[:4].cuda())
    assert rel(l4, logits[:4]) < 5e-3                               # samples do not see each other (bf16: B = 4 and B = 32 take different GEMM tiles)
    names = ["encoder.modality_encoder.language.encoder.layers.0.self_attn.q_proj.weight",
             "encoder.modality_encoder.language.encoder.layers.11.mlp.fc2.weight",
             "encoder.modality_encoder.image.encoder.layers.3.mlp.fc1.weight"]
    crit(logits, labels.cuda()).backward()
    g1 = {k: model.get_parameter(k).grad.clone() for k in names}
    tok1 = model.get_parameter("encoder.modality_encoder.language.embeddings.token_embedding.weight").grad.clone()
    model.zero_grad(set_to_none=True)
    (2.0 * crit(model(gdata, missing.cuda()), labels.cuda())).backward()
    for k in names:
        assert rel(model.get_parameter(k).grad, 2.0 * g1[k]) < 1e-3, k          # linear in the upstream gradient
    g2 = {k: model.get_parameter(k).grad.clone() for k in names}
    model.zero_grad(set_to_none=True)
    (2.0 * crit(model(gdata, missing.cuda()), labels.cuda())).backward()
    for k in names:
        assert torch.equal(model.get_parameter(k).grad, g2[k]), k               # ordered split-K reduce: bit-reproducible
    used = torch.zeros(49408, dtype=torch.bool)
    used[data["language"]["input_ids"].flatten()] = True
    assert float(tok1[(~used).cuda()].abs().max()) == 0.0 and float(tok1[used.cuda()].abs().max()) > 0.0


def test_config3_mixed_missing_codes_b32(pkg):
    """configs[3]: five modalities, B = 32, `synth_missing_index(32, mods, 0.3, 2025)` (the reference's 'mixed' recipe,
    src/utils/generate_missing.py:21-38, incl. this build's codes 5 = depth / 6 = thermal), bf16 instantiation.
    (a) every (sample, modality) whose code matches contributes exactly zero to that tower's embedding gradient, every other
    one a non-zero gradient; (b) the logits of 4 samples (chosen to cover codes 5 and 6) agree with the CPU oracle, fp32
    instantiation at 1e-3; (c) their gradients agree with the oracle's on the same 4-sample batch."""
    from missm_benchmark_amd.data import synth_missing_index
    from missm_benchmark_amd.nn import HipCrossEntropyLoss
    mods = ["video", "image", "audio", "depth", "thermal"]
    T = pkg.towers.TowerConfig
    cfgs = {m: T(kind="vision", add_time_attn=(m == "video"), num_frames=8 if m == "video" else 1) for m in mods}
    enc = pkg.lb.LanguageBind({m: f"LanguageBind_{m.capitalize()}" for m in mods}, configs=cfgs, compute_dtype=torch.bfloat16, seed=5)
    args = types.SimpleNamespace(modality_types=mods, feature_dims=768, fusion_dim=256, dropout_prob=0.0, fusion_type="sum")
    torch.manual_seed(0)
    model = pkg.base.finetune_model(args, 8, enc)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    B = 32
    missing = synth_missing_index(B, mods, 0.3, 2025)
    codes = {m: pkg.base.missing_type_index[m] for m in mods}
    assert int((missing != 0).sum()) == int(B * 0.3)
    assert {5, 6} <= set(missing.tolist()), "the seeded draw must exercise the extension codes (depth 5 / thermal 6)"
    g = torch.Generator().manual_seed(77)
    data = {m: {"pixel_values": torch.randn(*((B, 3, 8, 224, 224) if m == "video" else (B, 3, 224, 224)), generator=g)} for m in mods}
    labels = torch.randint(0, 8, (B,), generator=g)
    gdata = _to_gpu(data)
    emb = model.encoder(gdata)
    for e in emb.values():
        e.retain_grad()
    logits = model.fusion(emb, missing.cuda())
    HipCrossEntropyLoss()(logits, labels.cuda()).backward()
    for m in mods:
        ge = emb[m].grad
        hit = (missing == codes[m]).cuda()
        assert float(ge[hit].abs().max()) == 0.0 if bool(hit.any()) else True, m          # (a) exactly zero
        assert bool((ge[~hit].abs().amax(dim=1) > 0).all()), m
    assert all(model.get_parameter(f"encoder.modality_encoder.{m}.encoder.layers.0.mlp.fc1.weight").grad is not None for m in mods)
    # (b) + (c): 4 samples covering codes 5 and 6, a present-everything sample and one more missing code
    pick = [int((missing == 5).nonzero()[0]), int((missing == 6).nonzero()[0]), int((missing == 0).nonzero()[0])]
    others = [i for i in range(B) if int(missing[i]) not in (0, 5, 6)]
    pick.append(others[0] if others else int((missing == 0).nonzero()[1]))
    idx = torch.tensor(pick)
    tp, ocfg, proj, scales, fp = _oracle_parts(sd, mods)
    sub = {m: {"pixel_values": data[m]["pixel_values"][idx]} for m in mods}
    ologits, _ = O.finetune_forward(sub, missing[idx], tp, ocfg, proj, scales, fp, mods)
    oloss = O.cross_entropy(ologits, labels[idx])
    oloss.backward()
    assert rel(logits[idx.cuda()], ologits) < 5e-2                      # the B = 32 bf16 run, batch-independent rows
    model.encoder.set_compute_dtype(torch.float32)
    model.zero_grad(set_to_none=True)
    l4 = model(_to_gpu(sub), missing[idx].cuda())
    loss4 = HipCrossEntropyLoss()(l4, labels[idx].cuda())
    loss4.backward()
    assert rel(l4, ologits) < TOL32
    assert abs(float(loss4.detach()) - float(oloss.detach())) < TOL32 * max(1.0, float(oloss.detach()))
    checks = {"encoder.modality_encoder.depth.encoder.layers.0.self_attn.q_proj.weight": tp["depth"]["encoder.layers.0.self_attn.q_proj.weight"],
              "encoder.modality_encoder.thermal.encoder.layers.11.mlp.fc2.weight": tp["thermal"]["encoder.layers.11.mlp.fc2.weight"],
              "encoder.modality_encoder.video.encoder.layers.4.temporal_attn.out_proj.weight": tp["video"]["encoder.layers.4.temporal_attn.out_proj.weight"],
              "encoder.modality_proj.depth.weight": proj["depth"], "fusion.modal_proj.thermal.weight": fp["modal_proj.thermal.weight"],
              "fusion.modal_proj.depth.bias": fp["modal_proj.depth.bias"]}
    for k, ref in checks.items():
        assert grad_ok(k, model.get_parameter(k).grad, ref.grad, TOL32 * 4, torch.float32), k


def test_weights_changed_through_torch_reach_the_kernels(pkg):
    """after `.cuda()` the parameters are re-pointed views; writes through them (load_state_dict, a
    torch.optim step, p.mul_) must invalidate the bf16 weight copies the GEMMs read."""
    fix = load_golden("vision_tiny")
    x = fix["pixel_values"].cuda()
    tower = make_tower(pkg, fix["cfg"], "vision", fix["params"], torch.bfloat16)
    with torch.no_grad():
        _, p0 = tower(x)                                           # builds the bf16 shadows
    other = {k: v * 1.25 + 0.01 for k, v in fix["params"].items()}
    tower.load_state_dict(other)
    fresh = make_tower(pkg, fix["cfg"], "vision", other, torch.bfloat16)
    with torch.no_grad():
        _, p1 = tower(x)
        _, pf = fresh(x)
    assert torch.equal(p1, pf) and rel(p1, p0) > 1e-3              # the new weights are the ones that ran
    # torch.optim on the parameters (autograd gradient mode hands gradients to autograd like any module)
    pkg.towers.set_grad_mode("autograd")
    try:
        opt = torch.optim.SGD(tower.parameters(), lr=0.5)
        _, p = tower(x)
        p.square().sum().backward()
        opt.step()
        with torch.no_grad():
            _, p2 = tower(x)
        fresh2 = make_tower(pkg, fix["cfg"], "vision", {k: v.detach().cpu() for k, v in tower.state_dict().items()}, torch.bfloat16)
        with torch.no_grad():
            _, pf2 = fresh2(x)
        assert torch.equal(p2, pf2) and rel(p2, p1) > 1e-4
    finally:
        pkg.towers.set_grad_mode("direct")
    with torch.no_grad():
        tower.get_parameter("encoder.layers.0.mlp.fc1.weight").mul_(1.5)
        _, p3 = tower(x)
    assert rel(p3, p2) > 1e-4


def test_grad_accumulates_over_backwards_like_autograd(pkg):
    """two backward calls before the gradients are consumed ADD (micro-batches, a tower called twice); zero_grad resets"""
    fix = load_golden("video_tiny")
    tower = make_tower(pkg, fix["cfg"], "vision", fix["params"], torch.float32)
    x = fix["pixel_values"].cuda()
    cot = fix["cot_pooled"].cuda()
    names = [k for k in fix["grads"]]
    _, p = tower(x)
    (p * cot).sum().backward()
    g1 = {k: tower.get_parameter(k).grad.clone() for k in names}
    _, p = tower(x)
    (p * (0.5 * cot)).sum().backward()                              # second micro-batch: accumulates
    for k in names:
        if k.endswith("k_proj.bias"):
            continue
        assert rel(tower.get_parameter(k).grad, 1.5 * g1[k]) < 1e-4, k
    tower.zero_grad(set_to_none=True)
    _, p = tower(x)
    (p * cot).sum().backward()                                      # after zero_grad: a fresh gradient
    for k in names:
        if k.endswith("k_proj.bias"):
            continue
        assert rel(tower.get_parameter(k).grad, g1[k]) < 1e-5, k
    tower.zero_grad(set_to_none=False)                              # zeroed in place: accumulating into zeros is the same
    _, p = tower(x)
    (p * cot).sum().backward()
    for k in names:
        if k.endswith("k_proj.bias"):
            continue
        assert rel(tower.get_parameter(k).grad, g1[k]) < 1e-5, k


def test_eager_engine_rejects_second_backward(pkg):
    from missm_benchmark_amd.engine import TrainEngine
    fix = load_golden("vision_tiny")
    tower = make_tower(pkg, fix["cfg"], "vision", fix["params"], torch.float32)
    eng = TrainEngine(tower, lr=1e-3, eager_step=True)
    x = fix["pixel_values"].cuda()
    _, p = tower(x)
    p.sum().backward()
    _, p = tower(x)
    with pytest.raises(RuntimeError, match="second backward"):
        p.sum().backward()
    eng.step()
    _, p = tower(x)
    p.sum().backward()                                              # a new step is fine again
    eng.step()


def test_gradients_are_final_when_backward_returns(pkg):
    """``p.grad`` read on the caller's stream right after ``loss.backward()`` (no synchronize) is what it is after a device
    synchronize: the towers write their gradients on their own streams, and ``backward()`` must end with the caller's stream
    joined to them (a tower that runs alone on its stream used to race with such a read)."""
    from missm_benchmark_amd.nn import HipCrossEntropyLoss
    mods = ["image", "audio", "depth"]
    T = pkg.towers.TowerConfig
    enc = pkg.lb.LanguageBind({m: f"LanguageBind_{m.capitalize()}" for m in mods}, configs={m: T(kind="vision") for m in mods},
                              compute_dtype=torch.bfloat16, seed=5)
    args = types.SimpleNamespace(modality_types=mods, feature_dims=768, fusion_dim=256, dropout_prob=0.0, fusion_type="sum")
    model = pkg.base.finetune_model(args, 8, enc).cuda()
    g = torch.Generator().manual_seed(4)
    data = _to_gpu({m: {"pixel_values": torch.randn(16, 3, 224, 224, generator=g)} for m in mods})
    labels, missing = torch.randint(0, 8, (16,), generator=g).cuda(), torch.zeros(16, dtype=torch.int64).cuda()
    for grouped in (False, True):
        enc.group_towers = grouped
        model.zero_grad(set_to_none=True)
        HipCrossEntropyLoss()(model(data, missing), labels).backward()
        early = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        torch.cuda.synchronize()
        for k, p in model.named_parameters():
            if p.grad is not None:
                assert torch.equal(early[k], p.grad), (grouped, k)


def test_lockstep_tower_groups_equal_separate_towers(pkg):
    """image / audio / depth towers (one config, one input shape) run in lock-step with grouped GEMM launches (one tile grid for
    the three of them, forward, dX and dW); the result must be what the towers give one after the other - and what the oracle
    gives for two of the samples."""
    from missm_benchmark_amd.nn import HipCrossEntropyLoss
    mods = ["image", "audio", "depth"]
    T = pkg.towers.TowerConfig
    enc = pkg.lb.LanguageBind({m: f"LanguageBind_{m.capitalize()}" for m in mods}, configs={m: T(kind="vision") for m in mods},
                              compute_dtype=torch.bfloat16, seed=11)
    args = types.SimpleNamespace(modality_types=mods, feature_dims=768, fusion_dim=256, dropout_prob=0.0, fusion_type="sum")
    torch.manual_seed(0)
    model = pkg.base.finetune_model(args, 8, enc)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    B = 32
    g = torch.Generator().manual_seed(3)
    data = {m: {"pixel_values": torch.randn(B, 3, 224, 224, generator=g)} for m in mods}
    labels = torch.randint(0, 8, (B,), generator=g)
    missing = torch.zeros(B, dtype=torch.int64)
    missing[1], missing[5] = pkg.base.missing_type_index["audio"], pkg.base.missing_type_index["depth"]
    gdata = _to_gpu(data)
    names = [f"encoder.modality_encoder.{m}.{k}" for m in mods for k in
             ("encoder.layers.0.self_attn.q_proj.weight", "encoder.layers.7.mlp.fc1.weight", "encoder.layers.11.mlp.fc2.bias",
              "embeddings.patch_embedding.weight", "encoder.layers.3.layer_norm1.weight")]
    runs = {}
    for grouped in (True, False):
        enc.group_towers = grouped
        assert len(enc._units(gdata)) == (1 if grouped else 3)
        model.zero_grad(set_to_none=True)
        logits = model(gdata, missing.cuda())
        HipCrossEntropyLoss()(logits, labels.cuda()).backward()
        runs[grouped] = (logits.detach().clone(), {k: model.get_parameter(k).grad.clone() for k in names})
    assert rel(runs[True][0], runs[False][0]) < 2e-3
    for k in names:
        assert grad_ok(k, runs[True][1][k], runs[False][1][k], 1e-2, torch.bfloat16), k
    # two samples against the CPU oracle (batch-independent rows of the B = 32 lock-step run)
    idx = torch.tensor([0, 1])
    tp, ocfg, proj, scales, fp = _oracle_parts(sd, mods)
    with torch.no_grad():
        ologits, _ = O.finetune_forward({m: {"pixel_values": data[m]["pixel_values"][idx]} for m in mods}, missing[idx], tp, ocfg, proj, scales, fp, mods)
    assert rel(runs[True][0][:2], ologits) < 5e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_spectrogram_tower_with_more_than_256_tokens_vs_oracle(pkg, dtype):
    """a (112, 1040) spectrogram image is a 7 x 65 patch grid: 456 tokens per frame, more keys than the attention kernels hold in LDS
    at once - the key-chunked kernels (online softmax forward, chunked dQ / dK-dV backward) against the CPU oracle (the released
    audio checkpoint's 8 x 74 grid is the same code path)."""
    cfg = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, image_size=(112, 1040), patch_size=16)
    ocfg = O.VisionCfg(**cfg)
    params = O.init_tower_params(ocfg, 6)
    assert params["embeddings.position_embedding.weight"].shape[0] == 456
    tower = make_tower(pkg, cfg, "vision", params, dtype)
    x = torch.randn(2, 3, 112, 1040, generator=torch.Generator().manual_seed(8))
    last, pooled = tower(x.cuda())
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    olast, opooled = O.vision_tower(x, p, ocfg)
    tol = TOL32 if dtype == torch.float32 else 3e-2
    assert rel(pooled, opooled) < tol and rel(last, olast) < tol
    cot = torch.randn(pooled.shape, generator=torch.Generator().manual_seed(3))
    (pooled * cot.cuda()).sum().backward()
    (opooled * cot).sum().backward()
    for k in ("embeddings.position_embedding.weight", "encoder.layers.0.self_attn.q_proj.weight", "encoder.layers.0.self_attn.v_proj.weight",
              "encoder.layers.1.mlp.fc1.weight"):
        assert grad_ok(k, tower.get_parameter(k).grad, p[k].grad, 2e-3, dtype), k


@pytest.mark.parametrize("width", [(128, 256, 2), (1024, 4096, 16)], ids=["d128", "vit_l_width"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_patch14_tower_with_257_tokens_vs_oracle(pkg, dtype, width):
    """the geometry of the released ViT-L/14-class checkpoints: 14-pixel patches on a 224 x 224 image = 256 patches + CLS = 257 tokens.
    The patch matrix has 3 * 14 * 14 = 588 columns (rows padded to 592 in the flat store, the parameter is a strided view whose padding
    stays zero through Adam); 257 tokens take the key-chunked attention kernels.  Forward, gradients (patch embedding included) and one
    Adam step against the CPU oracle."""
    d, f, heads = width                 # (the second one is the ViT-L/14 layer: d = 1024, 16 heads of 64, MLP 4096; two layers of it)
    cfg = dict(hidden_size=d, intermediate_size=f, num_hidden_layers=2, num_attention_heads=heads, image_size=224, patch_size=14)
    ocfg = O.VisionCfg(**cfg)
    params = O.init_tower_params(ocfg, 9)
    assert params["embeddings.patch_embedding.weight"].shape == (d, 3, 14, 14)
    tower = make_tower(pkg, cfg, "vision", params, dtype)
    w = tower.get_parameter("embeddings.patch_embedding.weight")
    assert w.shape == (d, 3, 14, 14) and not w.is_contiguous() and torch.equal(w.detach().cpu(), params["embeddings.patch_embedding.weight"])
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(8))
    last, pooled = tower(x.cuda())
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    olast, opooled = O.vision_tower(x, p, ocfg)
    tol = TOL32 if dtype == torch.float32 else 3e-2
    assert rel(pooled, opooled) < tol and rel(last, olast) < tol
    cot = torch.randn(pooled.shape, generator=torch.Generator().manual_seed(3))
    (pooled * cot.cuda()).sum().backward()
    (opooled * cot).sum().backward()
    for k in ("embeddings.patch_embedding.weight", "embeddings.position_embedding.weight", "encoder.layers.0.self_attn.k_proj.weight",
              "encoder.layers.1.mlp.fc2.weight"):
        assert grad_ok(k, tower.get_parameter(k).grad, p[k].grad, 2e-3, dtype), k
    # one Adam step through the engine: the padded columns of the patch matrix stay exactly zero, the real ones move like torch's Adam
    from missm_benchmark_amd.engine import TrainEngine
    own_grad = w.grad.detach().cpu().clone()       # (the first Adam step is lr * sign(g): compare on the tower's own gradient)
    eng = TrainEngine(tower, lr=1e-3)
    eng.step()
    torch.cuda.synchronize()
    flat = tower.flat_master()
    b = tower._mat_blocks["patch"]
    blk = flat[b.offset:b.offset + b.numel].view(d, 592)
    assert float(blk[:, 588:].abs().max()) == 0.0
    ref = p["embeddings.patch_embedding.weight"].detach().clone().requires_grad_(True)
    ref.grad = own_grad
    torch.optim.Adam([ref], lr=1e-3).step()
    assert rel(w, ref.detach()) < 1e-5


def test_non_square_spectrogram_tower_vs_oracle(pkg):
    """the audio model's (num_mel_bins, target_length) image (reference resize_pos, image/modeling_image.py:795-839): a tower over a
    2 x 4 patch grid against the CPU oracle, forward and gradients, fp32 instantiation"""
    cfg = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, image_size=(32, 64), patch_size=16)
    ocfg = O.VisionCfg(**cfg)
    params = O.init_tower_params(ocfg, 4)
    assert params["embeddings.position_embedding.weight"].shape[0] == 9
    tower = make_tower(pkg, cfg, "vision", params, torch.float32)
    x = torch.randn(3, 3, 32, 64, generator=torch.Generator().manual_seed(2))
    last, pooled = tower(x.cuda())
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    olast, opooled = O.vision_tower(x, p, ocfg)
    assert rel(pooled, opooled) < TOL32 and rel(last, olast) < TOL32
    cot = torch.randn(pooled.shape, generator=torch.Generator().manual_seed(3))
    (pooled * cot.cuda()).sum().backward()
    (opooled * cot).sum().backward()
    for k in ("embeddings.position_embedding.weight", "embeddings.patch_embedding.weight", "encoder.layers.1.mlp.fc1.weight"):
        assert rel(tower.get_parameter(k).grad, p[k].grad) < 2e-3, k
    with pytest.raises(ValueError):
        tower(torch.randn(1, 3, 64, 32).cuda())               # height / width swapped


def _oracle_grad_lookup(tp, proj, fp):
    def get(k):
        if k.startswith("encoder.modality_encoder."):
            m = k.split(".")[2]
            return tp[m][k[len(f"encoder.modality_encoder.{m}."):]].grad
        if k.startswith("encoder.modality_proj."):
            return proj[k.split(".")[2]].grad
        return fp[k[len("fusion."):]].grad
    return get


def test_config2_full_batch_b32_vs_oracle_every_row(pkg):
    """BASELINE.json configs[2] / [3] at the size the metric is quoted on - five full ViT-B/16 towers, B = 32, `sum` fusion under
    `synth_missing_index(32, mods, 0.3, 2025)`, CE, full backward - against the CPU oracle on ALL 32 logits rows and on named
    gradients of every tower (VERDICT r2 #3c; the other B = 32 tests compare 4 rows).  The oracle differentiates the same batch in
    micro-batches of 4 samples (CE `sum` / 32: the samples are independent and the loss is their mean, so the accumulated gradient
    IS the full-batch gradient; 32 samples at once would hold ~60 GB of fp32 attention probabilities).
      fp32 instantiation: logits 1e-3, loss 1e-3, gradients 1e-3 in relative Frobenius norm (and the suite's 4e-3 worst-element bar);
      bf16 instantiation: logits 5e-2, gradients by direction, magnitude and the per-class Frobenius bound of `grad_ok`."""
    from missm_benchmark_amd.data import synth_missing_index
    from missm_benchmark_amd.nn import HipCrossEntropyLoss
    from test_towers_gpu import fro
    mods = ["video", "image", "audio", "depth", "thermal"]
    T = pkg.towers.TowerConfig
    cfgs = {m: T(kind="vision", add_time_attn=(m == "video"), num_frames=8 if m == "video" else 1) for m in mods}
    enc = pkg.lb.LanguageBind({m: f"LanguageBind_{m.capitalize()}" for m in mods}, configs=cfgs, compute_dtype=torch.float32, seed=9)
    args = types.SimpleNamespace(modality_types=mods, feature_dims=768, fusion_dim=256, dropout_prob=0.0, fusion_type="sum")
    torch.manual_seed(0)
    model = pkg.base.finetune_model(args, 8, enc)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    B, MB = 32, 4
    missing = synth_missing_index(B, mods, 0.3, 2025)
    g = torch.Generator().manual_seed(123)
    data = {m: {"pixel_values": torch.randn(*((B, 3, 8, 224, 224) if m == "video" else (B, 3, 224, 224)), generator=g)} for m in mods}
    labels = torch.randint(0, 8, (B,), generator=g)
    tp, ocfg, proj, scales, fp = _oracle_parts(sd, mods)
    rows, oloss = [], 0.0
    for lo in range(0, B, MB):
        sub = {m: {"pixel_values": data[m]["pixel_values"][lo:lo + MB]} for m in mods}
        lg, _ = O.finetune_forward(sub, missing[lo:lo + MB], tp, ocfg, proj, scales, fp, mods)
        part = torch.nn.functional.cross_entropy(lg, labels[lo:lo + MB], reduction="sum") / B
        part.backward()
        rows.append(lg.detach())
        oloss += float(part.detach())
    ologits = torch.cat(rows)
    oracle_grad = _oracle_grad_lookup(tp, proj, fp)
    per_tower = ["encoder.layers.0.self_attn.q_proj.weight", "encoder.layers.6.self_attn.k_proj.weight", "encoder.layers.11.self_attn.v_proj.weight",
                 "encoder.layers.5.self_attn.out_proj.weight", "encoder.layers.0.mlp.fc1.weight", "encoder.layers.11.mlp.fc2.weight",
                 "embeddings.patch_embedding.weight", "embeddings.position_embedding.weight", "embeddings.class_embedding",
                 "encoder.layers.3.layer_norm1.weight", "encoder.layers.9.layer_norm2.bias", "encoder.layers.2.mlp.fc1.bias",
                 "encoder.layers.7.self_attn.out_proj.bias", "pre_layrnorm.weight", "post_layernorm.bias"]
    video_only = ["encoder.layers.0.temporal_attn.q_proj.weight", "encoder.layers.8.temporal_attn.out_proj.weight",
                  "encoder.layers.4.temporal_embedding", "encoder.layers.10.temporal_layer_norm1.weight", "encoder.layers.1.temporal_attn.v_proj.bias"]
    names = [f"encoder.modality_encoder.{m}.{k}" for m in mods for k in per_tower] + \
            [f"encoder.modality_encoder.video.{k}" for k in video_only] + \
            [f"encoder.modality_proj.{m}.weight" for m in mods] + [f"fusion.modal_proj.{m}.weight" for m in mods] + \
            ["fusion.modal_proj.depth.bias", "fusion.norm.weight", "fusion.head.head.0.weight", "fusion.head.head.3.bias"]
    model = model.cuda()
    gdata = _to_gpu(data)
    for dtype, tol in ((torch.float32, TOL32), (torch.bfloat16, 5e-2)):
        enc.set_compute_dtype(dtype)
        model.zero_grad(set_to_none=True)
        logits = model(gdata, missing.cuda())
        loss = HipCrossEntropyLoss()(logits, labels.cuda())
        loss.backward()
        assert logits.shape == ologits.shape == (B, 8)
        assert rel(logits, ologits) < tol, dtype                               # every one of the 32 rows
        assert abs(float(loss.detach()) - oloss) < tol * max(1.0, oloss), dtype
        for k in names:
            mine, ref = model.get_parameter(k).grad, oracle_grad(k)
            assert mine is not None and ref is not None, k
            if dtype == torch.float32:
                assert fro(mine, ref) < 1e-3, (k, fro(mine, ref))
            assert grad_ok(k, mine, ref, tol * 4, dtype), (k, dtype)
        torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL32), (torch.bfloat16, TOLBF)])
def test_config4_video_tower_b16_vs_oracle(pkg, dtype, tol):
    """BASELINE.json configs[4] at its own shape: the video tower alone, B = 16 x 8 frames x 197 tokens (25 216 rows: the dispatcher
    takes the 256 x 128 kernel for the ungrouped products here, unlike the B = 32 / B = 2 runs - VERDICT r2 weak #3), forward +
    backward.  The cotangent is non-zero on two of the sixteen samples only: the weight gradients of the B = 16 run are then the
    oracle's gradients of those two samples (every other sample's rows enter the GEMMs with zero upstream gradient), the pooled
    outputs of the two are compared with the oracle directly, and the other fourteen through batch independence against a B = 2 run
    of the same tower (a different tile dispatch again)."""
    cfg = pkg.towers.TowerConfig(kind="vision", add_time_attn=True, num_frames=8)
    ocfg = O.VisionCfg(add_time_attn=True, num_frames=8)
    params = O.init_tower_params(ocfg, seed=31)
    tower = pkg.towers.ClipTower(cfg, compute_dtype=dtype)
    tower.load_state_dict(params, strict=True)
    tower = tower.cuda()
    B, pick = 16, [3, 11]
    x = torch.randn(B, 3, 8, 224, 224, generator=torch.Generator().manual_seed(17))
    cot_p = torch.zeros(B, 768)
    cot_p[pick] = torch.randn(2, 768, generator=torch.Generator().manual_seed(18))
    cot_l = torch.zeros(B * 8, 197, 768)
    for s in pick:                                                  # (the last hidden state's cotangent too, on the same two samples)
        cot_l[s * 8:(s + 1) * 8] = 0.05 * torch.randn(8, 197, 768, generator=torch.Generator().manual_seed(19 + s))
    op = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    olast, opooled = O.vision_tower(x[pick], op, ocfg)
    ((opooled * cot_p[pick]).sum() + (olast * torch.cat([cot_l[s * 8:(s + 1) * 8] for s in pick])).sum()).backward()
    last, pooled = tower(x.cuda())
    assert last.shape == (B * 8, 197, 768) and pooled.shape == (B, 768)
    ((pooled * cot_p.cuda()).sum() + (last * cot_l.cuda()).sum()).backward()
    assert rel(pooled[pick], opooled) < tol
    rows = torch.cat([torch.arange(s * 8, (s + 1) * 8) for s in pick])
    assert rel(last[rows.cuda()], olast) < tol
    names = ["encoder.layers.0.temporal_attn.q_proj.weight", "encoder.layers.11.temporal_attn.out_proj.weight", "encoder.layers.5.temporal_embedding",
             "encoder.layers.0.self_attn.v_proj.weight", "encoder.layers.6.self_attn.out_proj.weight", "encoder.layers.3.mlp.fc1.weight",
             "encoder.layers.11.mlp.fc2.weight", "encoder.layers.2.mlp.fc2.bias", "encoder.layers.9.layer_norm1.weight",
             "encoder.layers.4.temporal_layer_norm1.bias", "embeddings.patch_embedding.weight", "embeddings.position_embedding.weight",
             "pre_layrnorm.bias", "post_layernorm.weight"]
    for k in names:
        assert grad_ok(k, tower.get_parameter(k).grad, op[k].grad, tol * 4, dtype), k
    with torch.no_grad():                                           # batch independence of the other fourteen (B = 2 runs, another dispatch)
        for lo in (0, 8, 14):
            _, p2 = tower(x[lo:lo + 2].cuda())
            assert rel(p2, pooled[lo:lo + 2]) < (1e-4 if dtype == torch.float32 else 1e-2), lo
