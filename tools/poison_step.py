#!/usr/bin/env python3
"""Eager training steps with the allocator's free blocks poisoned (NaN) between and inside steps: a kernel that reads memory it has not written
this step shows up as NaN in the loss or in a gradient.  usage: poison_step.py [steps] [batch]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaot_amd import ops, _lib
from gaot_amd.trainer import TrainStep
dev = torch.device("cuda:0")
NSTEP = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
random.seed(0)
def poison(val=float("nan")):
    junk = [torch.full((random.randint(1, 256) * 16384,), val, device=dev) for _ in range(40)]
    junk += [torch.full((n,), val, device=dev) for n in (1 << 26, 1 << 25, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16, 4096, 1024, 256)]
    del junk
ops.register_grad_slots([], [])
torch.manual_seed(0)
model = bench.build_model().to(dev).train()
lat, x, p, t = bench.synthetic(1234, dev)
p, t = p[:B].contiguous(), t[:B].contiguous()
ts = TrainStep(model, lr=8e-4, weight_decay=1e-5, use_graph=False)
ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
names = [n for n, _ in model.named_parameters()]
for i in range(NSTEP):
    poison()
    l = float(ts.step())
    torch.cuda.synchronize()
    bad = [n for n, q in zip(names, model.parameters()) if not bool(torch.isfinite(q).all())]
    if l != l or bad:
        print("step", i, "loss", l, "non-finite parameters:", len(bad), bad[:8], flush=True)
        break
else:
    print("no NaN in", NSTEP, "poisoned steps; loss", l)
