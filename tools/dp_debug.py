import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.distributed as dist, torch.multiprocessing as mp
from test_dp_gpu import _model, _batch, KEYS

def one(model, eng, batch):
    from missm_benchmark_amd.nn import HipCrossEntropyLoss
    data, missing, labels = batch
    eng.zero_grad()
    loss = HipCrossEntropyLoss()(model({m: {k: v.cuda() for k, v in d.items()} for m, d in data.items()}, missing.cuda()), labels.cuda())
    loss.backward()

def worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from missm_benchmark_amd.engine import TrainEngine
    model = _model(seed=7 * (rank + 1)).cuda()
    eng = TrainEngine(model, lr=1e-3, eager_step=False, overlap=False)
    w0 = {k: model.get_parameter(k).detach().cpu().clone() for k in KEYS}
    one(model, eng, _batch(rank)); torch.cuda.synchronize()
    local = {k: model.get_parameter(k).grad.detach().cpu().clone() for k in KEYS}
    eng.reduce_gradients(); torch.cuda.synchronize()
    red = {k: model.get_parameter(k).grad.detach().cpu().clone() for k in KEYS}
    q.put((rank, w0, local, red))
    dist.destroy_process_group()

if __name__ == "__main__":
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, 2, 29711, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0]); [p.join() for p in ps]
    from missm_benchmark_amd.engine import TrainEngine
    model = _model(seed=7).cuda(); eng = TrainEngine(model, lr=1e-3, eager_step=False)
    ws = {k: model.get_parameter(k).detach().cpu().clone() for k in KEYS}
    for k in KEYS:
        print(k.split(".")[-3:], "w rank0==single", torch.equal(res[0][1][k], ws[k]), "w rank1==rank0", torch.equal(res[1][1][k], res[0][1][k]))
    for r in range(2):
        one(model, eng, _batch(r)); torch.cuda.synchronize()
        for k in KEYS:
            g = model.get_parameter(k).grad.detach().cpu()
            print(" rank", r, k.split(".")[-2:], "local vs single:", float((res[r][2][k] - g).abs().max() / g.abs().max()))
    for k in KEYS:
        s = res[0][2][k] + res[1][2][k]
        print(" sum(local) vs reduced", k.split(".")[-2:], float((s - res[0][3][k]).abs().max() / s.abs().max()))
