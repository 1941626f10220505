#!/usr/bin/env python3
"""A few training steps of a BASELINE configuration for the profiler.  usage: eager_steps.py c2|c3|c4 N [graph] [staged]
c2 = the bench configuration; c4 = its first 4 samples (the 4 096-token batch of BASELINE configs[3]); c3 = NACA-shaped skewed meshes, vx mode, batch 16, 8192 nodes (tests/_workloads.naca_points)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
import torch
import bench
from gaot_amd.trainer import TrainStep
from gaot_amd.model.gaot import GAOT
from gaot_amd.model.layers.magno import MAGNOConfig
from gaot_amd.model.layers.attn import TransformerConfig
from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch
from tests._workloads import grid, naca_points

which, n = sys.argv[1], int(sys.argv[2])
graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
dev = torch.device("cuda:0")
torch.manual_seed(0)
if which in ("c2", "c4"):
    model = bench.build_model().to(dev).train()
    lat, x, p, t = bench.synthetic(1234, dev)
    kw = dict(latent_tokens_coord=lat, xcoord=x)
    B = bench.BATCH
    if which == "c4":
        B = 4
        p, t = p[:4].contiguous(), t[:4].contiguous()
else:
    B, N = 16, 8192
    mc = MAGNOConfig(radius=0.033, lifting_channels=64, precompute_edges=True)
    model = GAOT(3, 1, NS(args=NS(magno=mc, transformer=TransformerConfig(patch_size=2, hidden_size=256)), latent_tokens_size=[64, 64])).to(dev).train()
    g = torch.Generator().manual_seed(0)
    lat = grid([64, 64]).to(dev)
    x = torch.stack([naca_points(N, g, 0.15) for _ in range(B)]).to(dev)
    ns = NeighborSearch("native")
    enc = [[ns(x[b], lat, 0.033)] for b in range(B)]
    dec = [[ns(lat, x[b], 0.033)] for b in range(B)]
    p, t = torch.randn(B, N, 3, device=dev), torch.randn(B, N, 1, device=dev)
    kw = dict(latent_tokens_coord=lat, xcoord=x, encoder_nbrs=enc, decoder_nbrs=dec)
ts = TrainStep(model, use_graph=graph, staged=("staged" in sys.argv) or None)
ts.bind(p, t, **kw)
for _ in range(3):
    ts.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    ts.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(json.dumps({"config": which, "ms_per_step": 1e3 * dt, "samples_per_s": B / dt, "hipgraph": graph}))
