"""Data-parallel engine on the GPU: 2 ranks (gloo, both on device 0 - RCCL refuses two ranks on one device) must
(a) stay bit-identical to each other, (b) match one process training on the concatenated batch (1e-4) and the CPU oracle's
gradient of that batch (1e-3), and (c) give the same parameters on the path bench.py / train() actually run - bucketed
all-reduce overlapped with the backward, Adam enqueued on the communication stream behind each tower's reduction
(eager_step=True, overlap=True) - as on the plain path (reduce everything, then step)."""
import os
import sys
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(seed=0, language=False):
    torch.manual_seed(1000 + seed)   # the fusion head draws from torch's global RNG (like the reference's nn.Linear init)
    sys.path.insert(0, ROOT)
    import missm_benchmark_amd as M
    lb, base = M.install()
    from missm_benchmark_amd.towers import TowerConfig
    tiny = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, image_size=32, patch_size=16)
    cfgs = {"image": TowerConfig(kind="vision", **tiny), "video": TowerConfig(kind="vision", add_time_attn=True, num_frames=4, **tiny)}
    tcfg = TowerConfig(kind="text", hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=2, vocab_size=64,
                       max_position_embeddings=8)
    enc = lb.LanguageBind({"image": "i", "video": "v"}, configs=cfgs, text_config=tcfg, projection_dim=32, compute_dtype=torch.float32,
                          seed=seed)
    mods = (["language"] if language else []) + ["image", "video"]
    args = types.SimpleNamespace(modality_types=mods, feature_dims=32, fusion_dim=16, dropout_prob=0.0, fusion_type="sum")
    return base.finetune_model(args, 3, enc)


def _batch(rank, B=4, language=False):
    g = torch.Generator().manual_seed(50 + rank)
    data = {"image": {"pixel_values": torch.randn(B, 3, 32, 32, generator=g)},
            "video": {"pixel_values": torch.randn(B, 3, 4, 32, 32, generator=g)}}
    if language:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import missm_oracle as O
        ids, mask = O.synth_text_batch(B, 8, 90 + rank, vocab=64)
        data = {"language": {"input_ids": ids, "attention_mask": mask}, **data}
    missing = torch.tensor([0, 4, 2, 0])[:B]
    labels = torch.randint(0, 3, (B,), generator=g)
    return data, missing, labels


KEYS = ["encoder.modality_encoder.video.encoder.layers.0.temporal_attn.q_proj.weight",
        "encoder.modality_encoder.image.encoder.layers.1.mlp.fc2.weight", "encoder.modality_proj.image.weight",
        "fusion.head.head.3.weight", "encoder.modality_encoder.video.embeddings.position_embedding.weight"]


def _train(model, batches, steps=2, lr=1e-3, eager=False):
    """returns (mean gradients of the first step as the optimizer sees them, parameters after `steps` steps).
    eager=True is the default path of bench.py / train(): `loss.backward()` all-reduces the gradient buckets as they become
    final and applies each tower's Adam on the communication stream; gradients cannot be observed before the update there."""
    from missm_benchmark_amd.engine import TrainEngine
    from missm_benchmark_amd.nn import HipCrossEntropyLoss
    eng = TrainEngine(model, lr=lr, eager_step=eager, overlap=True)
    crit = HipCrossEntropyLoss()
    grads = {}
    for it in range(steps):
        data, missing, labels = batches
        eng.zero_grad()
        loss = crit(model({m: {k: v.cuda() for k, v in d.items()} for m, d in data.items()}, missing.cuda()), labels.cuda())
        loss.backward()
        if eager:
            eng.step()
            continue
        eng.reduce_gradients()
        if it == 0:
            torch.cuda.synchronize()
            grads = {k: (model.get_parameter(k).grad.detach() / eng.world).cpu().clone() for k in KEYS}
        eng.apply_adam()
    torch.cuda.synchronize()
    return grads, {k: model.get_parameter(k).detach().cpu().clone() for k in KEYS}


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        model = _model(seed=7 * (rank + 1)).cuda()          # different init per rank: the engine must broadcast rank 0's
        out = _train(model, _batch(rank))
        model = _model(seed=7 * (rank + 1)).cuda()
        _, eager_params = _train(model, _batch(rank), eager=True)
        # by value (numpy), not as shared-memory tensors: the parent may fetch the item after this process has exited
        q.put((rank, tuple({k: v.numpy() for k, v in d.items()} for d in out + (eager_params,)), ""))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, None, traceback.format_exc()))


def test_two_ranks_match_single_process_on_concatenated_batch():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[2] == "", r[2]
    res = [(r[0], tuple({k: torch.from_numpy(v) for k, v in d.items()} for d in r[1]), r[2]) for r in res]
    for k in KEYS:
        assert torch.equal(res[0][1][1][k], res[1][1][1][k]), f"ranks diverged on {k}"
        assert torch.equal(res[0][1][0][k], res[1][1][0][k]), f"reduced gradients differ on {k}"
    # single process, global batch = both ranks' samples (mean loss over 8 = mean of the two per-rank means):
    # the all-reduced mean gradient must equal the single-process gradient (Adam's sign-like first steps would amplify
    # rounding noise on near-zero gradients, so gradients - not parameters - are compared)
    d0, m0, l0 = _batch(0)
    d1, m1, l1 = _batch(1)
    data = {m: {"pixel_values": torch.cat([d0[m]["pixel_values"], d1[m]["pixel_values"]])} for m in d0}
    sgrads, _ = _train(_model(seed=7).cuda(), (data, torch.cat([m0, m1]), torch.cat([l0, l1])), steps=1)
    for k, v in sgrads.items():
        err = float((v - res[0][1][0][k]).abs().max() / v.abs().max())
        assert err < 1e-4, (k, err)
    # the default path (eager Adam behind the bucketed, overlapped all-reduce) lands on the same parameters as the plain one
    # (not bit-for-bit: bias / LayerNorm gradients are summed with fp32 atomics, whose arrival order differs from run to run;
    #  a last-bit difference in step 1 reaches every weight in step 2.  Two Adam steps of 1e-3 move a weight by ~2e-3.)
    for r in res:
        for k in KEYS:
            d = float((r[1][2][k] - r[1][1][k]).abs().max())
            assert d < 2e-6, f"rank {r[0]}: eager/overlap path diverged from the plain path on {k}: {d:.3e}"
    for k in KEYS:      # the two ranks of the eager run share every reduced value: bit-identical
        assert torch.equal(res[0][1][2][k], res[1][1][2][k]), f"eager path: ranks diverged on {k}"
    # ... and the reduced gradient is the CPU oracle's gradient of the concatenated batch (rank 0's initial parameters)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import missm_oracle as O
    m0_ = _model(seed=7)
    sd = {k: v.detach().clone() for k, v in m0_.state_dict().items()}
    tiny = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2, image_size=32, patch_size=16)
    ocfg = {"image": O.VisionCfg(**tiny), "video": O.VisionCfg(add_time_attn=True, num_frames=4, **tiny)}
    tp = {m: {k[len(f"encoder.modality_encoder.{m}."):]: v.requires_grad_(True) for k, v in sd.items()
              if k.startswith(f"encoder.modality_encoder.{m}.")} for m in ocfg}
    proj = {m: sd[f"encoder.modality_proj.{m}.weight"].requires_grad_(True) for m in ocfg}
    fp = {k[len("fusion."):]: v.requires_grad_(True) for k, v in sd.items() if k.startswith("fusion.")}
    ologits, _ = O.finetune_forward(data, torch.cat([m0, m1]), tp, ocfg, proj, {m: torch.tensor(2.6592) for m in ocfg}, fp,
                                    ["image", "video"])
    O.cross_entropy(ologits, torch.cat([l0, l1])).backward()

    def ograd(k):
        if k.startswith("encoder.modality_encoder."):
            m = k.split(".")[2]
            return tp[m][k[len(f"encoder.modality_encoder.{m}."):]].grad
        return proj[k.split(".")[2]].grad if k.startswith("encoder.modality_proj.") else fp[k[len("fusion."):]].grad

    for k in KEYS:
        ref = ograd(k)
        err = float((res[0][1][0][k] - ref).abs().max() / ref.abs().max())
        assert err < 2e-3, (k, err)


# ---------------------------------------------------------------------------------------------------------------------------
# The reference's own wrap (train_ddp.py:188-189): SyncBatchNorm.convert_sync_batchnorm + DistributedDataParallel(device_ids=
# [local_rank], broadcast_buffers=True, find_unused_parameters=False) around the drop-in model, gradients handed to autograd
# (towers.set_grad_mode("autograd")) - against the bundled engine on the same two ranks (VERDICT r2 #6 / missing #3).
# ---------------------------------------------------------------------------------------------------------------------------
DDP_KEYS = KEYS + ["encoder.modality_encoder.language.embeddings.token_embedding.weight",
                   "encoder.modality_encoder.language.encoder.layers.0.self_attn.v_proj.weight",
                   "encoder.modality_encoder.image.pre_layrnorm.bias", "fusion.modal_proj.language.bias"]


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from torch.nn.parallel import DistributedDataParallel as DDP
        model = _model(seed=7 * (rank + 1), language=True).cuda()     # different init per rank: DDP's constructor broadcasts rank 0's
        from missm_benchmark_amd import towers
        from missm_benchmark_amd.engine import TrainEngine
        from missm_benchmark_amd.nn import HipCrossEntropyLoss
        crit = HipCrossEntropyLoss()
        data, missing, labels = _batch(rank, language=True)
        gdata = {m: {k: v.cuda() for k, v in d.items()} for m, d in data.items()}
        towers.set_grad_mode("autograd")
        ddp = DDP(torch.nn.SyncBatchNorm.convert_sync_batchnorm(model), device_ids=[0], broadcast_buffers=True, find_unused_parameters=False)
        for _ in range(2):       # a second iteration: where a registered parameter without a gradient would make the reducer raise
            ddp.zero_grad(set_to_none=True)
            crit(ddp(gdata, missing.cuda()), labels.cuda()).backward()
        torch.cuda.synchronize()
        g_ddp = {k: ddp.module.get_parameter(k).grad.detach().cpu().numpy() for k in DDP_KEYS}
        n_grad = sum(p.grad is not None for p in ddp.module.parameters()), sum(1 for _ in ddp.module.parameters())
        towers.set_grad_mode("direct")
        model2 = _model(seed=7 * (rank + 1), language=True).cuda()
        eng = TrainEngine(model2, lr=1e-3, overlap=True)              # broadcasts rank 0's parameters, like the DDP constructor
        eng.zero_grad()
        crit(model2(gdata, missing.cuda()), labels.cuda()).backward()
        eng.reduce_gradients()
        torch.cuda.synchronize()
        g_eng = {k: (model2.get_parameter(k).grad.detach() / eng.world).cpu().numpy() for k in DDP_KEYS}
        same_init = all(bool(torch.equal(a.detach().cpu(), b.detach().cpu())) for a, b in zip(ddp.module.parameters(), model2.parameters()))
        q.put((rank, (g_ddp, g_eng, n_grad, same_init), ""))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, None, traceback.format_exc()))


def test_distributed_data_parallel_wrap_equals_engine():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + (os.getpid() % 300)
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[2] == "", r[2]
    for rank, (g_ddp, g_eng, (with_grad, total), same_init), _ in res:
        assert with_grad == total, f"rank {rank}: {total - with_grad} registered parameters got no gradient through DDP"
        assert same_init, "DDP's and the engine's broadcast left different parameters"
        for k in DDP_KEYS:
            a, b = torch.from_numpy(g_ddp[k]), torch.from_numpy(g_eng[k])
            err = float((a - b).abs().max() / b.abs().max())
            assert err < 1e-4, (rank, k, err)                          # DDP's averaged gradient = the engine's reduced mean gradient
    for k in DDP_KEYS:                                                 # and both ranks hold the same reduced values
        assert (res[0][1][0][k] == res[1][1][0][k]).all(), k
