import os, sys
sys.path.insert(0, "/root/repo")
import torch
import bench
from gaot_amd import ops, _lib
from gaot_amd.trainer import TrainStep
dev = torch.device("cuda:0")
lib = _lib.load()
def make():
    ops.register_grad_slots([], [])
    torch.manual_seed(0)
    model = bench.build_model().to(dev).train()
    lat, x, p, t = bench.synthetic(1234, dev)
    p, t = p[:4].contiguous(), t[:4].contiguous()
    ts = TrainStep(model, lr=8e-4, weight_decay=1e-5, use_graph=False)
    ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
    return ts, model
a, ma = make(); b, mb = make()
names = [n for n, _ in ma.named_parameters()]
for i in range(80):
    lib.gaot_debug_split_redo_count(1)
    la = a.step(); torch.cuda.synchronize(); ra = lib.gaot_debug_split_redo_count(1)
    lb = b.step(); torch.cuda.synchronize(); rb = lib.gaot_debug_split_redo_count(1)
    ga = a.bucket.flat.clone() if hasattr(a, "bucket") else None
    gb = b.bucket.flat.clone() if hasattr(b, "bucket") else None
    same_w = [torch.equal(p.detach(), q.detach()) for p, q in zip(ma.parameters(), mb.parameters())]
    if not all(same_w) or ra != rb or ra != 0:
        bad = [n for n, s in zip(names, same_w) if not s]
        print("step", i, "redo", ra, rb, "loss", float(la), float(lb), "differing params", len(bad), bad[:6], flush=True)
        if not all(same_w): break
print("done")
