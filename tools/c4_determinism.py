#!/usr/bin/env python3
"""Two identical TrainSteps of the 4 096-token configuration (the attention backward shares query tiles between two workgroups and adds their
dK / dV by atomics): after N steps the losses and the weights must agree bit for bit, graph and eager."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaot_amd import ops, _lib
from gaot_amd.trainer import TrainStep
dev = torch.device("cuda:0")
def make(graph):
    ops.register_grad_slots([], [])
    torch.manual_seed(0)
    model = bench.build_model().to(dev).train()
    lat, x, p, t = bench.synthetic(1234, dev)
    p, t = p[:4].contiguous(), t[:4].contiguous()
    ts = TrainStep(model, lr=8e-4, weight_decay=1e-5, use_graph=graph)
    ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
    return ts, model
lib = _lib.load()
if len(sys.argv) > 1: lib.gaot_debug_set_attention_qsplit(int(sys.argv[1]))
for graph in (True, False):
    runs = []
    for rep in range(2):
        ts, model = make(graph)
        losses = []
        for i in range(60):
            l = ts.step() if not graph else (ts.step(), ts._loss)[1]
            if i % 20 == 19: losses.append(float(l))
        torch.cuda.synchronize()
        runs.append((losses, [q.detach().clone() for q in model.parameters()]))
    same = all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1]))
    print("graph" if graph else "eager", "losses", runs[0][0], runs[1][0], "weights identical:", same, flush=True)
