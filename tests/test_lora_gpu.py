"""Adapter-only LoRA training of the vision towers (reference languagebind/image/modeling_image.py:775-793, configuration_image.py:200-202:
``get_peft_model(vision_model.encoder, LoraConfig(r, lora_alpha, target_modules))`` - frozen encoder, trainable rank-r adapters; embeddings
and the two outer LayerNorms stay trainable) on the HIP path, against the CPU oracle's unmerged peft forward (oracle.linear; peft itself is
absent and unpinned upstream: the wrapper's arithmetic is restated from its published definition - PARITY UNPINNED for that piece, the
tower underneath is pinned by the reference fixtures).

  * state-dict keys are peft's (``encoder.base_model.model.layers.N.self_attn.q_proj.{base_layer.weight, lora_A.default.weight, ...}``);
  * step 0 identity: the LoRA tower computes what a plain tower with the merged weights W + (alpha / r) B A computes;
  * forward, and the gradients of every trainable tensor, against the oracle (fp32 1e-3; bf16 by its own bars); frozen tensors: no grad;
  * one Adam step through the engine: adapters / embeddings move as the oracle's Adam moves them, every frozen tensor is bit-identical,
    and the next forward uses the updated adapters (the merged GEMM weights are rebuilt);
  * gradients accumulate over two backward passes like autograd's.
"""
import pytest
import torch

import missm_oracle as O
from test_towers_gpu import TOL32, TOLBF, grad_ok, pkg, rel  # noqa: F401  (pkg is a fixture)

pytestmark = pytest.mark.gpu

TINY = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, image_size=32, patch_size=16)
CASES = {
    "image": dict(),                                                              # targets: self_attn q/k/v/out
    "video_time": dict(add_time_attn=True, num_frames=4),                         # targets: temporal_attn q/k/v/out
    "image_time_mlp": dict(add_time_attn=True, num_frames=4, temporal_mlp=True),  # + temporal_mlp.fc1 / fc2
}


def _targets(cfg):
    out = []
    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{i}"
        attn = "temporal_attn" if cfg.add_time_attn else "self_attn"
        out += [(f"{p}.{attn}.{n}_proj", cfg.hidden_size, cfg.hidden_size) for n in ("k", "v", "q", "out")]
        if cfg.add_time_attn and cfg.temporal_mlp:
            out += [(f"{p}.temporal_mlp.fc1", cfg.intermediate_size, cfg.hidden_size), (f"{p}.temporal_mlp.fc2", cfg.hidden_size, cfg.intermediate_size)]
    return out


def _params(cfg, seed):
    """oracle-keyed parameters: the seeded base tower + adapters with NON-zero B (peft starts B at zero; here the branch must matter)"""
    p = O.init_tower_params(cfg, seed)
    g = torch.Generator().manual_seed(seed + 1)
    for stem, n_out, k_in in _targets(cfg):
        p[stem + ".lora_A.default.weight"] = torch.randn(cfg.lora_r, k_in, generator=g) * 0.2
        p[stem + ".lora_B.default.weight"] = torch.randn(n_out, cfg.lora_r, generator=g) * 0.02      # (alpha / r) B A ~ the base weight's size
    return p


def _peft_keys(p, cfg):
    """the keys peft gives the wrapped encoder (written here independently of the tower's own mapping)"""
    stems = {t[0] for t in _targets(cfg)}
    out = {}
    for k, v in p.items():
        if k.startswith("encoder.layers."):
            stem, leaf = k.rsplit(".", 1)
            rest = k[len("encoder."):]
            if stem in stems and ".lora_" not in k:
                rest = stem[len("encoder."):] + ".base_layer." + leaf
            k = "encoder.base_model.model." + rest
        out[k] = v
    return out


def _merged(p, cfg):
    s = cfg.lora_alpha / cfg.lora_r
    out = {k: v.clone() for k, v in p.items() if ".lora_" not in k}
    for stem, _, _ in _targets(cfg):
        out[stem + ".weight"] = out[stem + ".weight"] + s * p[stem + ".lora_B.default.weight"] @ p[stem + ".lora_A.default.weight"]
    return out


def _inputs(cfg, B, seed):
    g = torch.Generator().manual_seed(seed)
    shape = (B, 3, cfg.num_frames, 32, 32) if cfg.num_frames > 1 else (B, 3, 32, 32)
    return torch.randn(*shape, generator=g)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, TOL32), (torch.bfloat16, TOLBF)])
@pytest.mark.parametrize("case", list(CASES))
def test_lora_tower_forward_backward_vs_oracle(pkg, case, dtype, tol):
    T = pkg.towers.TowerConfig
    ocfg = O.VisionCfg(**TINY, lora_r=2, lora_alpha=16.0, **CASES[case])
    p = _params(ocfg, 41)
    tower = pkg.towers.ClipTower(T(kind="vision", **TINY, lora_r=2, lora_alpha=16.0, **CASES[case]), compute_dtype=dtype)
    assert set(tower.state_dict()) == set(_peft_keys(p, ocfg))
    tower.load_state_dict(_peft_keys(p, ocfg), strict=True)
    tower = tower.cuda()
    trainable = {k for k, q in tower.named_parameters() if q.requires_grad}
    assert trainable == {k for k in _peft_keys(p, ocfg) if ".lora_" in k or not k.startswith("encoder.")}
    x = _inputs(ocfg, 3, 5)
    op = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    olast, opooled = O.vision_tower(x, op, ocfg)
    last, pooled = tower(x.cuda())
    assert rel(pooled, opooled) < tol and rel(last, olast) < tol
    # step-0 identity with the merged model: a plain tower holding W + (alpha / r) B A computes the same function
    plain_cfg = {k: v for k, v in {**TINY, **CASES[case]}.items()}
    plain = pkg.towers.ClipTower(T(kind="vision", **plain_cfg), compute_dtype=dtype)
    plain.load_state_dict(_merged(p, ocfg), strict=True)
    with torch.no_grad():
        _, pm = plain.cuda()(x.cuda())
    assert rel(pooled, pm) < (1e-5 if dtype == torch.float32 else 2e-2)
    cp = torch.randn(pooled.shape, generator=torch.Generator().manual_seed(7))
    ch = torch.randn(last.shape, generator=torch.Generator().manual_seed(8)) * 0.1
    ((opooled * cp).sum() + (olast * ch).sum()).backward()
    ((pooled * cp.cuda()).sum() + (last * ch.cuda()).sum()).backward()
    orig = _peft_keys({k: k for k in p}, ocfg)          # peft key -> the oracle's key
    for k, q in tower.named_parameters():
        if k in trainable:
            assert q.grad is not None, k
            assert grad_ok(orig[k], q.grad, op[orig[k]].grad, tol * 4, dtype, fro_scale=2.0), k
        else:
            assert q.grad is None, f"frozen {k} got a gradient"
    # a second backward before the gradients are consumed ADDS (autograd semantics), also for dA / dB derived from the scratch dW
    g1 = {k: q.grad.clone() for k, q in tower.named_parameters() if k in trainable}
    last, pooled = tower(x.cuda())
    ((pooled * cp.cuda()).sum() + (last * ch.cuda()).sum()).backward()
    for k, q in tower.named_parameters():
        if k in trainable and not k.endswith("k_proj.bias"):
            assert rel(q.grad, 2.0 * g1[k]) < (1e-4 if dtype == torch.float32 else 2e-2), k


@pytest.mark.parametrize("case", ["image", "image_time_mlp"])
def test_lora_adam_step_moves_only_the_trainable_tensors(pkg, case):
    from missm_benchmark_amd.engine import TrainEngine
    T = pkg.towers.TowerConfig
    ocfg = O.VisionCfg(**TINY, lora_r=2, lora_alpha=16.0, **CASES[case])
    p = _params(ocfg, 43)
    tower = pkg.towers.ClipTower(T(kind="vision", **TINY, lora_r=2, lora_alpha=16.0, **CASES[case]), compute_dtype=torch.float32)
    tower.load_state_dict(_peft_keys(p, ocfg), strict=True)
    tower = tower.cuda()
    pub = {v: k for k, v in _peft_keys({k: k for k in p}, ocfg).items()}          # the oracle's key -> peft key
    x = _inputs(ocfg, 3, 9)
    # (a large cotangent: Adam's first step is lr * g / (|g| + eps) - keep every |g| far above eps = 1e-8, where rounding would decide the sign)
    cp = 1e3 * torch.randn(3, 64, generator=torch.Generator().manual_seed(10))
    eng = TrainEngine(tower, lr=1e-2)
    before = {k: v.detach().clone() for k, v in tower.state_dict().items()}
    eng.zero_grad()
    _, pooled = tower(x.cuda())
    (pooled * cp.cuda()).sum().backward()
    eng.step()
    # oracle: the same gradient, one torch.optim.Adam step on the trainable tensors only
    op = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    _, opooled = O.vision_tower(x, op, ocfg)
    (opooled * cp).sum().backward()
    after = tower.state_dict()
    moved = 0
    for k, v in p.items():
        mine = after[pub[k]].cpu()
        if ".lora_" in k or not k.startswith("encoder."):
            want = v.clone()
            O.adam_step(want, op[k].grad, torch.zeros_like(v), torch.zeros_like(v), 1, 1e-2)
            assert rel(mine, want) < 1e-4, k
            moved += int(not torch.equal(mine, before[pub[k]].cpu()))
        else:
            assert torch.equal(mine, before[pub[k]].cpu()), f"frozen {k} changed"
    assert moved >= 4 * ocfg.num_hidden_layers * 2
    # the next forward runs on the UPDATED adapters (merged GEMM weights rebuilt) = the oracle with the updated parameters
    with torch.no_grad():
        _, p2 = tower(x.cuda())
        _, o2 = O.vision_tower(x, {k: after[pub[k]].cpu() for k in p}, ocfg)
    assert rel(p2, o2) < TOL32 and rel(p2, pooled) > 1e-3
    sd = eng.state_dict()           # optimizer state: trainable tensors only, like torch's Adam over requires_grad parameters
    names = sd["param_groups"][0]["param_names"]
    assert all((".lora_" in names[i]) or not names[i].startswith("encoder.") for i in sd["state"]) and len(sd["state"]) == moved


def test_lora_model_in_finetune_step(pkg):
    """a LoRA image tower next to a plain (fully trained) video tower inside finetune_model under the engine's eager path: the step
    runs, the LoRA tower's frozen encoder stays put, its adapters move, the plain tower trains as before"""
    import types
    from missm_benchmark_amd.engine import TrainEngine
    from missm_benchmark_amd.nn import HipCrossEntropyLoss
    T = pkg.towers.TowerConfig
    cfgs = {"image": T(kind="vision", **TINY, lora_r=2), "video": T(kind="vision", add_time_attn=True, num_frames=4, **TINY)}
    tcfg = T(kind="text", hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2, vocab_size=64, max_position_embeddings=8)
    enc = pkg.lb.LanguageBind({"image": "i", "video": "v"}, configs=cfgs, text_config=tcfg, projection_dim=32, compute_dtype=torch.float32, seed=3)
    with torch.no_grad():       # peft starts lora_B at zero (gradient into A vanishes at step 0): make the adapters live
        for k, q in enc.modality_encoder["image"].named_parameters():
            if ".lora_B." in k:
                q.normal_(0.0, 0.1)
    args = types.SimpleNamespace(modality_types=["image", "video"], feature_dims=32, fusion_dim=16, dropout_prob=0.0, fusion_type="sum")
    model = pkg.base.finetune_model(args, 3, enc).cuda()
    eng = TrainEngine(model, lr=1e-3, eager_step=True)
    g = torch.Generator().manual_seed(2)
    data = {"image": {"pixel_values": torch.randn(4, 3, 32, 32, generator=g).cuda()},
            "video": {"pixel_values": torch.randn(4, 3, 4, 32, 32, generator=g).cuda()}}
    labels, missing = torch.randint(0, 3, (4,), generator=g).cuda(), torch.tensor([0, 4, 2, 0]).cuda()
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    losses = []
    for _ in range(3):
        eng.zero_grad()
        loss = HipCrossEntropyLoss()(model(data, missing), labels)
        loss.backward()
        eng.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]
    for k, v in model.state_dict().items():
        same = torch.equal(v, before[k])
        if k.startswith("encoder.modality_encoder.image.encoder.") and ".lora_" not in k:
            assert same, f"frozen {k} changed"
        elif k.startswith("encoder.modality_encoder.image.") or k.startswith("encoder.modality_encoder.video."):
            if not k.endswith("k_proj.bias") and "position_ids" not in k:
                assert not same, f"{k} did not train"
