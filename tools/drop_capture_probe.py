#!/usr/bin/env python3
"""bisect: which part of a device-side edge drop does not survive a hipGraph capture"""
import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
import torch
from gaot_amd import plan as P, ops, _lib as L
from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch
from tests._workloads import grid, naca_points
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = naca_points(4096, g, 0.2).to(dev)
lat = grid([64, 64]).to(dev)
nb = NeighborSearch("native")(x, lat, 0.033)
base = P.plan_for(nb, x.shape[0])
print("E", base.E, "Q", base.Q, flush=True)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
for mode, kw in (("max_neighbors", dict(max_neighbors=6)), ("ratio", dict(sample_ratio=0.6))):
    dp = P.dropped_plan(base, mode, **kw)
    torch.cuda.synchronize()
    print(mode, "eager kept", int(dp.e_dev.item()), flush=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    steps = {"seed": lambda: L.check(L.load().gaot_attention_seed_next(ops._p(ops.dropout_state(dev)), 1, ops._p(dp._seed), ops._stream())),
             "redraw": dp.redraw,
             "feat": lambda: (dp.redraw(), dp.edge_features(x, lat)),
             "cos": lambda: (dp.redraw(), dp.cosine_attention(x, lat)),
             "stats": lambda: (dp.redraw(), dp.geo_stats(x, lat)),
             "all": lambda: (dp.redraw(), dp.edge_features(x, lat), dp.cosine_attention(x, lat), dp.geo_stats(x, lat), dp.inv_deg_edge)}
    for name, fn in steps.items():
        if which not in ("all", name) and not (which == "each"):
            continue
        with torch.cuda.stream(s):
            fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        print("capturing", mode, name, flush=True)
        with torch.cuda.graph(gr, stream=s, capture_error_mode="thread_local"):
            fn()
        print("captured", flush=True)
        ks = []
        for _ in range(3):
            gr.replay()
            torch.cuda.synchronize()
            ks.append(int(dp.e_dev.item()))
        print(mode, name, "replays kept", ks, flush=True)
