#!/usr/bin/env python3
"""Soak: N TrainStep steps of the bench configuration (hipGraph) + N eager ones; loss finite and falling, allocator footprint flat.
usage: soak.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaot_amd.trainer import TrainStep

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0")
for graph in (True, False):
    torch.manual_seed(0)
    model = bench.build_model().to(dev).train()
    lat, x, p, t = bench.synthetic(1234, dev)
    ts = TrainStep(model, use_graph=graph)
    ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
    l0 = float(ts.step())
    torch.cuda.synchronize()
    m0 = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved())
    steps = n if graph else max(20, n // 20)
    for _ in range(steps):
        loss = ts.step()
    torch.cuda.synchronize()
    m1 = (torch.cuda.memory_allocated(), torch.cuda.memory_reserved())
    l1 = float(loss)
    print(f"graph={graph}: {steps} steps, loss {l0:.6f} -> {l1:.6f}, step counter {float(ts.opt.step_count):.0f}, "
          f"allocated {m0[0] >> 20} -> {m1[0] >> 20} MiB, reserved {m0[1] >> 20} -> {m1[1] >> 20} MiB")
    assert l1 == l1 and l1 < l0 and float(ts.opt.step_count) == steps + 1
    assert m1[1] <= m0[1] + (64 << 20), "allocator footprint grew"
    del ts, model
    torch.cuda.empty_cache()
