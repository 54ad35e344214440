"""Is the two-tower (text + image) step bit-reproducible?  forward logits and weight gradients across identical calls,
with the towers on two streams and on one."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch
import missm_benchmark_amd as M
import missm_oracle as O
from missm_benchmark_amd.nn import HipCrossEntropyLoss
from missm_benchmark_amd.towers import TowerConfig as T

lb, base = M.install()
enc = lb.LanguageBind({"image": "i"}, configs={"image": T(kind="vision")}, text_config=T(kind="text"), compute_dtype=torch.bfloat16, seed=3)
args = types.SimpleNamespace(modality_types=["language", "image"], feature_dims=768, fusion_dim=256, dropout_prob=0.0, fusion_type="sum")
torch.manual_seed(0)
model = base.finetune_model(args, 8, enc).cuda()
B = 32
g = torch.Generator().manual_seed(40)
ids, mask = O.synth_text_batch(B, 77, 41)
data = {"language": {"input_ids": ids.cuda(), "attention_mask": mask.cuda()}, "image": {"pixel_values": torch.randn(B, 3, 224, 224, generator=g).cuda()}}
labels = torch.randint(0, 8, (B,), generator=g).cuda()
missing = torch.zeros(B, dtype=torch.int64).cuda()
crit = HipCrossEntropyLoss()
names = [n for n, _ in model.named_parameters() if any(s in n for s in ("layers.0.self_attn.q_proj.weight", "layers.11.mlp.fc2.weight", "layers.0.layer_norm1.weight", "layers.5.mlp.fc1.bias", "modality_proj", "head.head.3.weight", "position_embedding"))]
for par in (True, False):
    enc.parallel_streams = par
    runs = []
    for it in range(3):
        model.zero_grad(set_to_none=True)
        emb = model.encoder(data)
        for e in emb.values():
            e.retain_grad()
        logits = model.fusion(emb, missing)
        (crit(logits, labels) * (2.0 if it == 2 else 1.0)).backward()
        torch.cuda.synchronize()
        runs.append((logits.detach().clone(), {m: e.grad.clone() for m, e in emb.items()}, {n: model.get_parameter(n).grad.clone() for n in names}))
    print(f"parallel_streams={par}: logits equal {torch.equal(runs[0][0], runs[1][0])}")
    for m in runs[0][1]:
        print(f"   d emb[{m}] equal {torch.equal(runs[0][1][m], runs[1][1][m])}; x2 exact {torch.equal(2 * runs[0][1][m], runs[2][1][m])}")
    for n in names:
        a, b, c = runs[0][2][n], runs[1][2][n], runs[2][2][n]
        r = float((a - b).abs().max() / a.abs().max())
        r2 = float((2 * a - c).abs().max() / a.abs().max())
        print(f"   {n[-60:]:60s} same-call diff {r:.2e}   x2 diff {r2:.2e}")
