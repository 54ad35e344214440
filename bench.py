#!/usr/bin/env python3
"""Benchmark of the MissM-Benchmark hot path on MI355X: multimodal samples/s, fwd + bwd + all-reduce + Adam.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
           bench.py --gpus N --steps K --warmup W          (N > 1: one rank per GPU, RCCL)

Workload (BASELINE.json configs[2], the config the metric is quoted on): 5 modalities (video T=8, image, audio, depth,
thermal), synthetic N(0,1) pixels, B = 32 per GPU, ViT-B/16 towers (d=768, L=12, 197 tokens), `sum` fusion, C = 8
classes, random-init weights, bf16 GEMM/attention operands with fp32 accumulation, residual stream, master weights and
Adam.  A step = forward, cross-entropy, full backward through all five towers, gradient all-reduce (N > 1), fused Adam
on every parameter that received a gradient.  Weak scaling: per-GPU batch fixed.

The JSON line also carries
  roofline     : the dominant kernel (bf16 MFMA GEMM): algorithmic FLOPs of every launch in the timed region / their
                 summed durations from HIP events recorded on the launching stream, against the 2.5 PFLOP/s dense bf16 peak
  cpu_baseline : the CPU oracle (the validated restatement of the reference) timed on this box's host cores on a bounded
                 sample of the same workload (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # entry point: kernel arguments staged in device memory (missm_benchmark_amd/__init__.py); before torch loads HIP; the effective value is recorded in the JSON line

MODALITIES = ["image", "audio", "depth", "thermal", "video"]   # video last in forward => first in backward (largest all-reduce overlaps the rest)
PEAK_BF16_TFLOPS = 2500.0   # dense, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3     # fp32-input MFMA (= the fp32 vector rate), same guide
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="samples per GPU")
    ap.add_argument("--missing", type=float, default=0.0, help="missing-modality ratio (configs[3] uses 0.3)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--modalities", default=",".join(MODALITIES))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2, help="samples per CPU-baseline pass (2 warm-ups + median of 3 passes)")
    ap.add_argument("--no-fp32-line", action="store_true", help="skip the short fp32-instantiation pass after the timed region")
    ap.add_argument("--gemm-table", action="store_true", help="print per-shape GEMM time/TFLOP/s of the roofline pass to stderr")
    ap.add_argument("--serial-streams", action="store_true", help="encode the modalities on one stream (per-kernel timings are then exclusive)")
    return ap.parse_args()


PMC_RECORD = os.path.join("profiles", "r03_pmc_hbm_traffic.json")


def pmc_traffic(family, launches_per_step):
    """HBM-side bytes per launch of one kernel family from the committed PMC passes (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    runs of this script with --serial-streams, FETCH_SIZE doubled per the gfx950 correction; tools/pmc_traffic.py).  This is a RECORD
    of a separate profiling pass, not something this run measured: it is printed with the commit and the date of that pass
    (VERDICT r2 #4), and it is None when the record is absent or was taken on another workload."""
    try:
        d = json.load(open(os.path.join(ROOT, PMC_RECORD)))
        k = d["kernels"][family]
        return {"bytes_per_launch": int(k["bytes_per_step"] / launches_per_step), "bytes_per_step": int(k["bytes_per_step"]), "unit": "B",
                "source": PMC_RECORD, "pmc_pass_commit": d.get("commit", "unknown"), "pmc_pass_date": d.get("date", "unknown"),
                "note": "recorded by a separate rocprofv3 --pmc pass of this script (not measured in this run)"}
    except Exception:
        return None


def workload_name(modalities, args) -> str:
    ms = set(modalities)
    if ms == set(MODALITIES):
        return "configs[3]: 5-modality with %d%% missing-modality codes" % round(args.missing * 100) if args.missing > 0 else \
               "configs[2]: 5-modality (video T=8/image/audio/depth/thermal)"
    if ms == {"language", "image"}:
        return "configs[1]: image + text two-modality fusion"
    if ms == {"video"}:
        return "configs[4]: video tower (8 frames x 197 tokens, factorised time attention)"
    return "custom: " + "+".join(modalities)


def usable_cores() -> int:
    """cores this process may actually use: affinity mask capped by the cgroup CPU quota (a GPU box hands out a share of
    the host; sizing the thread pool by os.cpu_count() would oversubscribe it many times over)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(modalities, cpu_batch, warmups=2, passes=3):
    """Time the CPU oracle (oracle/missm_oracle.py, the validated restatement of the reference: kind "port") on a bounded
    sample of the same workload: fwd + CE + bwd of `cpu_batch` samples, `warmups` untimed passes, median of `passes`
    (SURVEY.md 8d / BASELINE.md 2), on every core this process may use."""
    import statistics
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import missm_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfgs, params, proj, scales = {}, {}, {}, {}
    for i, m in enumerate(modalities):
        if m == "language":
            cfgs[m] = O.TextCfg()
            params[m] = {k: v.requires_grad_(True) for k, v in O.init_tower_params(cfgs[m], seed=i, kind="text").items()}
        else:
            cfgs[m] = O.VisionCfg(add_time_attn=(m == "video"), num_frames=8 if m == "video" else 1)
            params[m] = {k: v.requires_grad_(True) for k, v in O.init_tower_params(cfgs[m], seed=i).items()}
            scales[m] = torch.tensor(2.6592)
        proj[m] = (torch.randn(768, 768, generator=torch.Generator().manual_seed(100 + i)) * 768 ** -0.5).requires_grad_(True)
    fp = {k: v.requires_grad_(True) for k, v in O.init_fusion_params(modalities, 768, 256, 8, seed=7).items()}
    g = torch.Generator().manual_seed(1)
    data = {}
    for m in modalities:
        if m == "language":
            ids, mask = O.synth_text_batch(cpu_batch, 77, 100)
            data[m] = {"input_ids": ids, "attention_mask": mask}
        else:
            data[m] = {"pixel_values": torch.randn(*((cpu_batch, 3, 8, 224, 224) if m == "video" else (cpu_batch, 3, 224, 224)), generator=g)}
    missing = torch.zeros(cpu_batch, dtype=torch.int64)
    labels = torch.randint(0, 8, (cpu_batch,), generator=g)
    leaves = [t for d in params.values() for t in d.values()] + list(proj.values()) + list(fp.values())

    def one():
        for t in leaves:
            t.grad = None
        logits, _ = O.finetune_forward(data, missing, params, cfgs, proj, scales, fp, modalities)
        O.cross_entropy(logits, labels).backward()

    print(f"[bench] cpu_baseline: oracle fwd+bwd of {cpu_batch} sample(s) x {len(modalities)} modalities on {cores} host threads: "
          f"{warmups} warm-up + {passes} timed passes ...", file=sys.stderr, flush=True)
    for _ in range(warmups):
        one()
    times = []
    for _ in range(passes):
        t0 = time.perf_counter()
        one()
        times.append(time.perf_counter() - t0)
        print(f"[bench] cpu_baseline: pass {len(times)}: {times[-1]:.2f} s", file=sys.stderr, flush=True)
    dt = statistics.median(times)
    return {"value": round(cpu_batch / dt, 4), "unit": "samples/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model_name(), "torch": torch.__version__, "threads": torch.get_num_threads(),
            "sample": f"{cpu_batch} sample(s) x {len(modalities)} modalities ({'+'.join(modalities)}), fwd+CE+bwd (no optimizer), fp32 torch "
                      f"CPU oracle; median of {passes} passes ({', '.join(f'{t:.2f}' for t in times)} s) after {warmups} warm-ups"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU product path)"
    torch.cuda.set_device(local if local < torch.cuda.device_count() else 0)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  MISSM_DIST_BACKEND=gloo lets the N > 1 engine path be rehearsed with several ranks on ONE GPU
        # (RCCL refuses two ranks on one device); ranks then share device 0.
        dist.init_process_group(os.environ.get("MISSM_DIST_BACKEND", "nccl"), init_method="env://")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    import missm_benchmark_amd as M
    from missm_benchmark_amd import ops
    from missm_benchmark_amd.engine import TrainEngine
    from missm_benchmark_amd.nn import HipCrossEntropyLoss
    lb, base = M.install()
    modalities = args.modalities.split(",")
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    clip_type = {m: f"LanguageBind_{m.capitalize()}" for m in modalities if m != "language"}
    enc = lb.LanguageBind(clip_type, compute_dtype=cdt, seed=0, allow_synthetic=True)   # random-init ViT-B/16-class towers (no checkpoints offline)
    enc.parallel_streams = not args.serial_streams
    margs = types.SimpleNamespace(modality_types=modalities, feature_dims=768, fusion_dim=256, dropout_prob=0.1, fusion_type="sum")
    model = base.finetune_model(margs, 8, enc).cuda()
    model.train()
    engine = TrainEngine(model, lr=1e-4, eager_step=True)   # one backward per step: Adam rides right behind each tower's gradient
    criterion = HipCrossEntropyLoss()
    B = args.batch
    g = torch.Generator().manual_seed(1 + rank)
    data = {}
    for m in modalities:
        if m == "language":
            from missm_benchmark_amd.data import synth_text_batch
            ids, mask = synth_text_batch(B, 77, seed=100 + rank)
            data[m] = {"input_ids": ids.cuda(), "attention_mask": mask.cuda()}
            continue
        shape = (B, 3, 8, 224, 224) if m == "video" else (B, 3, 224, 224)
        data[m] = {"pixel_values": torch.randn(*shape, generator=g).cuda()}
    labels = torch.randint(0, 8, (B,), generator=g).cuda()
    if args.missing > 0:
        from missm_benchmark_amd.data import synth_missing_index
        missing = synth_missing_index(B, modalities, args.missing, 2025 + rank).cuda()
    else:
        missing = torch.zeros(B, dtype=torch.int64, device="cuda")

    def step():
        engine.zero_grad()
        logits = model(data, missing)
        loss = criterion(logits, labels)
        loss.backward()
        engine.step()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    # Per-launch HIP-event timing is only meaningful when kernels do not overlap: with one stream per tower (default) the
    # roofline pass is a single-stream replay of the same step right after the timed region; with --serial-streams the
    # events are recorded inside the timed region itself.
    inline_prof = args.serial_streams and not args.no_roofline
    prof = [] if inline_prof else None
    ops.GEMM_PROFILE = prof
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_ms = (time.perf_counter() - t0) / args.steps * 1e3     # the host's share: time to ENQUEUE a step (before the closing barrier)
    barrier()
    dt = time.perf_counter() - t0
    ops.GEMM_PROFILE = None
    tmax = torch.tensor([dt], device="cuda")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax)
    final_loss = float(loss.detach())
    replay_ms = None
    if not args.no_roofline and not inline_prof:
        enc.parallel_streams = False
        step(); barrier()
        prof = []
        aprof = []
        ops.GEMM_PROFILE = prof
        ops.ATTN_PROFILE = aprof
        t1 = time.perf_counter()
        for _ in range(2):
            step()
        barrier()
        replay_ms = (time.perf_counter() - t1) / 2 * 1e3
        ops.GEMM_PROFILE = None
        ops.ATTN_PROFILE = None
        enc.parallel_streams = True

    # The parity gate (1e-3 vs the reference's fp32 arithmetic) is met by the fp32 instantiation of the same kernel source.  It is the
    # reference-precision number, so it gets a real measurement: 3 warm-ups + 10 timed steps with the same two-stream schedule, and its
    # own roofline from a single-stream replay (GEMM flops over the summed launch times against the 157.3 TFLOP/s fp32 matrix peak).
    fp32_line = None
    if args.dtype == "bf16" and not args.no_fp32_line and world == 1:
        F32_WARMUP, F32_STEPS = 3, 10
        enc.set_compute_dtype(torch.float32)
        for _ in range(F32_WARMUP):
            step()
        barrier()
        t2 = time.perf_counter()
        for _ in range(F32_STEPS):
            step()
        barrier()
        f32_ms = (time.perf_counter() - t2) / F32_STEPS * 1e3
        fp32_line = {"samples_per_s": round(B * world / (f32_ms * 1e-3), 2), "ms_per_step": round(f32_ms, 2), "steps": F32_STEPS, "warmup": F32_WARMUP,
                     "dtype": "f32 (v_mfma_f32_16x16x4_f32 GEMM/attention operands; the instantiation held to the 1e-3 parity gate)"}
        if not args.no_roofline:
            enc.parallel_streams = False
            step(); barrier()
            fprof = []
            ops.GEMM_PROFILE = fprof
            t3 = time.perf_counter()
            for _ in range(2):
                step()
            barrier()
            f_replay = (time.perf_counter() - t3) / 2 * 1e3
            ops.GEMM_PROFILE = None
            enc.parallel_streams = not args.serial_streams
            fl = sum(p[2] for p in fprof)
            fms = sum(p[0].elapsed_time(p[1]) for p in fprof)
            fach = fl / (fms * 1e-3) / 1e12
            fp32_line["roofline"] = {"kernel": "gemm_kernel<float,*> (v_mfma_f32_16x16x4_f32): the GEMM family of the fp32 instantiation", "bound": "mfma",
                                     "achieved": round(fach, 1), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(fach / PEAK_F32_TFLOPS, 4),
                                     "traffic": None, "launches": len(fprof), "avg_launch_us": round(fms * 1e3 / len(fprof), 2),
                                     "measured_on": "single-stream replay of the same fp32 step", "gemm_ms_per_step": round(fms / 2, 2),
                                     "serial_ms_per_step": round(f_replay, 2),
                                     "whole_step_TFLOP_per_s": round(fl / 2 / (f32_ms * 1e-3) / 1e12, 1)}
        enc.set_compute_dtype(torch.bfloat16)
        step(); barrier()

    roof_attn = None
    if not args.no_roofline and not inline_prof and aprof:
        # the north star's second figure: the attention kernels (QK^T / softmax / PV and their backward) - stand-alone they are
        # HBM-bound (98.5 FLOP/B against a ridge of 310), so the bound is HBM; the MFMA fraction is reported next to it
        ams = sum(p[0].elapsed_time(p[1]) for p in aprof)
        afl, aby = sum(p[2] for p in aprof), sum(p[3] for p in aprof)
        roof_attn = {"kernel": "attn_fwd_mfma_kernel / attn_bwd_mfma_kernel (all attention launches of a step)", "bound": "hbm",
                     "achieved": round(aby / (ams * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(aby / (ams * 1e-3) / 1e9 / 8000.0, 4),
                     "traffic": pmc_traffic("attn", len(aprof) / 2) if (set(modalities) == set(MODALITIES) and B == 32 and args.dtype == "bf16") else None,
                     "mfma_TFLOP_per_s": round(afl / (ams * 1e-3) / 1e12, 1), "mfma_frac": round(afl / (ams * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                     "launches": len(aprof), "attention_ms_per_step": round(ams / 2, 2)}
    if roof_attn is not None and prof:
        # the north star words the attention block as "QKV projection + softmax.V": the fused-QKV projection GEMMs (N = 3K forward,
        # K = 3N / M = 3N for its dX / dW) and the attention kernels TOGETHER, flops over the sum of their launch times.  The QKV
        # projection stays a separate GEMM launch (training must write qkv for the backward either way - DESIGN 4.2).
        is_qkv_f = lambda sh: sh[3] == 0 and sh[4] == 0 and sh[1] == 3 * sh[2]          # noqa: E731  [rows, 3d, d]
        is_qkv_b = lambda sh: (sh[3] == 0 and sh[4] == 0 and sh[2] == 3 * sh[1]) or (sh[3] == 1 and sh[0] == 3 * sh[1])   # noqa: E731
        qf = [(p[0].elapsed_time(p[1]), p[2]) for p in prof if is_qkv_f(p[3])]
        qb = [(p[0].elapsed_time(p[1]), p[2]) for p in prof if is_qkv_b(p[3])]
        af = [(p[0].elapsed_time(p[1]), p[2]) for p in aprof if p[4] == "fwd"]
        ab = [(p[0].elapsed_time(p[1]), p[2]) for p in aprof if p[4] == "bwd"]
        tf = lambda items: sum(f for _, f in items) / (sum(t for t, _ in items) * 1e-3) / 1e12 if items else 0.0   # noqa: E731
        roof_attn["qkv_projection_plus_attention"] = {
            "forward_TFLOP_per_s": round(tf(qf + af), 1), "forward_mfma_frac": round(tf(qf + af) / PEAK_BF16_TFLOPS, 4),
            "forward_backward_TFLOP_per_s": round(tf(qf + af + qb + ab), 1),
            "forward_backward_mfma_frac": round(tf(qf + af + qb + ab) / PEAK_BF16_TFLOPS, 4),
            "qkv_gemm_forward_TFLOP_per_s": round(tf(qf), 1), "attention_forward_TFLOP_per_s": round(tf(af), 1),
            "attention_backward_TFLOP_per_s": round(tf(ab), 1),
            # the ceiling of a stand-alone attention kernel: q, k, v in + o out = 100.9 KB for 9.935 MFLOP per (frame, head) = 98.5 FLOP/B
            #  times the HBM rate the kernels actually reach - what the attention GEMMs could do at most at that rate
            "attention_forward_bound": {"flop_per_byte": 98.5, "measured_hbm_GB_per_s": round(sum(p[3] for p in aprof if p[4] == "fwd") / (sum(t for t, _ in af) * 1e-3) / 1e9, 1) if af else None,
                                        "bound_TFLOP_per_s": round(98.5 * sum(p[3] for p in aprof if p[4] == "fwd") / (sum(t for t, _ in af) * 1e-3) / 1e12, 1) if af else None,
                                        "bound_at_8TBs_TFLOP_per_s": 788.0, "bound_at_8TBs_mfma_frac": 0.3152}}
    roof = None
    if prof:
        flops = sum(p[2] for p in prof)
        ms = sum(p[0].elapsed_time(p[1]) for p in prof)
        if args.gemm_table and rank == 0:
            agg = {}
            for e0, e1, f, shape in prof:
                a = agg.setdefault(shape, [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += f
            for shape, (n, t, f) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print(f"[gemm] M,N,K,tA,tB,act={shape}: {n:4d} launches {t:8.2f} ms  {f / t / 1e9:7.1f} TFLOP/s", file=sys.stderr)
        ach = flops / (ms * 1e-3) / 1e12
        default_workload = set(modalities) == set(MODALITIES) and B == 32 and args.dtype == "bf16"   # what the PMC passes ran
        roof = {"kernel": "gemm8p_kernel / gemm8p_tn_kernel (+ gemm4w_kernel on row remainders, gemm_kernel<bf16,..> on small shapes): the MFMA GEMM family behind every linear, dX and dW" if args.dtype == "bf16" else "gemm_kernel<float,*>", "bound": "mfma",
                "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3, "unit": "TFLOP/s",
                "frac": round(ach / (PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3), 4),
                "traffic": pmc_traffic("gemm", len(prof) / (args.steps if inline_prof else 2)) if default_workload else None,
                "launches": len(prof), "avg_launch_us": round(ms * 1e3 / len(prof), 2),
                "measured_on": "timed region (single stream)" if inline_prof else "single-stream replay of the same step after the timed region",
                "gemm_ms_per_step": round(ms / (args.steps if inline_prof else 2), 2),
                "serial_ms_per_step": round(dt / args.steps * 1e3, 2) if inline_prof else round(replay_ms, 2)}
    if rank == 0:
        out = {"metric": "multimodal samples/sec (fwd+bwd) at B=32, 5 modalities", "value": round(B * world * args.steps / dt, 2),
               "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 2), "host_enqueue_ms_per_step": round(host_ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": workload_name(modalities, args) + ": ViT-B/16 towers + sum fusion, fwd+bwd+allreduce+Adam", "per_gpu_batch": B, "global_batch": B * world,
                          "modalities": modalities, "missing_ratio": args.missing, "params": engine.num_parameters(),
                          "parallelism": f"dp{world}", "final_loss": round(final_loss, 4),
                          "HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG", "unset"),
                          # the loss reads the pooled output only: the last layer's out-projection / MLP (forward and backward) run on the
                          # CLS rows, whose results are the only ones that reach the loss or any gradient (0 = all rows)
                          "last_layer_on_cls_rows": os.environ.get("MISSM_SPARSE_LAST", "1") != "0"},
               "roofline": roof}
        if roof_attn is not None:
            out["roofline_attention"] = roof_attn
        if fp32_line is not None:
            out["fp32_instantiation"] = fp32_line
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(modalities, args.cpu_batch)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
