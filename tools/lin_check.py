import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_dp_gpu import _model, _batch, KEYS
from missm_benchmark_amd.nn import HipCrossEntropyLoss
def grads(model, batch):
    data, missing, labels = batch
    for p in model.parameters(): p.grad = None
    loss = HipCrossEntropyLoss()(model({m: {k: v.cuda() for k, v in d.items()} for m, d in data.items()}, missing.cuda()), labels.cuda())
    loss.backward(); torch.cuda.synchronize()
    return {k: model.get_parameter(k).grad.detach().cpu().clone() for k in KEYS}, float(loss)
model = _model(seed=7).cuda()
d0, m0, l0 = _batch(0); d1, m1, l1 = _batch(1)
g0, L0 = grads(model, (d0, m0, l0)); g1, L1 = grads(model, (d1, m1, l1))
data = {m: {"pixel_values": torch.cat([d0[m]["pixel_values"], d1[m]["pixel_values"]])} for m in d0}
gc, Lc = grads(model, (data, torch.cat([m0, m1]), torch.cat([l0, l1])))
print("loss", L0, L1, (L0+L1)/2, Lc)
for k in KEYS:
    avg = (g0[k] + g1[k]) / 2
    print(k, float((avg - gc[k]).abs().max() / gc[k].abs().max()))
