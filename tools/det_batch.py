#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaot_amd import ops, _lib
from gaot_amd.trainer import TrainStep
dev = torch.device("cuda:0")
lib = _lib.load()
NSTEP = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
def make(graph):
    ops.register_grad_slots([], [])
    torch.manual_seed(0)
    model = bench.build_model().to(dev).train()
    lat, x, p, t = bench.synthetic(1234, dev)
    p, t = p[:B].contiguous(), t[:B].contiguous()
    ts = TrainStep(model, lr=8e-4, weight_decay=1e-5, use_graph=graph)
    ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
    return ts, model
for graph in (True, False):
    runs = []
    for rep in range(2):
        ts, model = make(graph)
        l = None; hist = []
        for i in range(NSTEP):
            l = ts.step() if not graph else (ts.step(), ts._loss)[1]
            if not graph: hist.append(float(l))
        torch.cuda.synchronize()
        runs.append((float(l), [q.detach().clone() for q in model.parameters()], hist))
    same = all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1]))
    if not graph and not same:
        d = [i for i, (u, v) in enumerate(zip(runs[0][2], runs[1][2])) if u != v]
        print("first differing loss at step", d[0] if d else None, runs[0][2][d[0]] if d else None, runs[1][2][d[0]] if d else None, flush=True)
    print(os.path.basename(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "B", B, "graph" if graph else "eager", "last losses", runs[0][0], runs[1][0], "weights identical:", same, flush=True)
