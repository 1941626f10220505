#!/usr/bin/env python3
"""Every torch.empty / empty_like / new_empty on the device comes back filled with NaN (floats) or 0x7f7f7f7f (ints): a kernel that reads an
output or a workspace before it has written it shows up as NaN in the loss or the weights.  usage: nan_empty.py [steps] [batch] [graph]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
_e, _el, _ne = torch.empty, torch.empty_like, torch.Tensor.new_empty
FILL = float(sys.argv[3]) if len(sys.argv) > 3 else float('nan')          # (a huge finite value finds what max / compare paths swallow as NaN)
def _fill(t):
    if t.is_cuda and t.numel():
        if t.dtype.is_floating_point: t.fill_(FILL)
        elif t.dtype in (torch.int32, torch.int64): t.fill_(0x7f7f7f7f)
    return t
torch.empty = lambda *a, **k: _fill(_e(*a, **k))
torch.empty_like = lambda *a, **k: _fill(_el(*a, **k))
torch.Tensor.new_empty = lambda self, *a, **k: _fill(_ne(self, *a, **k))
import bench
from gaot_amd import ops, _lib
from gaot_amd.trainer import TrainStep
dev = torch.device("cuda:0")
NSTEP = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ops.register_grad_slots([], [])
torch.manual_seed(0)
model = bench.build_model().to(dev).train()
lat, x, p, t = bench.synthetic(1234, dev)
p, t = p[:B].contiguous(), t[:B].contiguous()
ts = TrainStep(model, lr=8e-4, weight_decay=1e-5, use_graph=False)
ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
names = [n for n, _ in model.named_parameters()]
for i in range(NSTEP):
    l = float(ts.step())
    torch.cuda.synchronize()
    bad = [n for n, q in zip(names, model.parameters()) if not bool(torch.isfinite(q).all())]
    print("step", i, "loss", repr(l), "redone tiles", int(_lib.load().gaot_debug_split_redo_count(1)), "non-finite parameters:", len(bad), bad[:6], flush=True)
    if l != l or bad: break
