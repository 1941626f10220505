#!/usr/bin/env python3
"""same-box A/B of an integer debug switch of the library on C3 (NACA-shaped vx meshes, 16 x 8 192 nodes, one batch bound and replayed):
    python tools/c3_ab.py gaot_debug_set_wgrad_slab_rule 0 1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
import torch
from gaot_amd import ops, _lib
from gaot_amd.trainer import TrainStep
from gaot_amd.model.gaot import GAOT
from gaot_amd.model.layers.magno import MAGNOConfig
from gaot_amd.model.layers.attn import TransformerConfig
from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch
from tests._workloads import grid, naca_points
lib = _lib.load()
setter = getattr(lib, sys.argv[1])
vals = [int(v) for v in sys.argv[2:]]
dev = torch.device("cuda:0")
B, N = 16, 8192
g = torch.Generator().manual_seed(0)
lat = grid([64, 64]).to(dev)
x = torch.stack([naca_points(N, g, 0.15) for _ in range(B)]).to(dev)
ns = NeighborSearch("native")
enc = [[ns(x[b], lat, 0.033)] for b in range(B)]
dec = [[ns(lat, x[b], 0.033)] for b in range(B)]
p, t = torch.randn(B, N, 3, device=dev), torch.randn(B, N, 1, device=dev)
res = {}
for v in vals:
    old = setter(v)
    ops._PATH_CACHE.clear(); ops.register_grad_slots([], [])
    torch.manual_seed(0)
    mc = MAGNOConfig(radius=0.033, lifting_channels=64, precompute_edges=True)
    model = GAOT(3, 1, NS(args=NS(magno=mc, transformer=TransformerConfig(patch_size=2, hidden_size=256)), latent_tokens_size=[64, 64])).to(dev).train()
    ts = TrainStep(model, use_graph=True)
    ts.bind(p, t, latent_tokens_coord=lat, xcoord=x, encoder_nbrs=enc, decoder_nbrs=dec)
    for _ in range(6):
        ts.step()
    torch.cuda.synchronize()
    res[v] = ts
    setter(old)
for rnd in range(3):
    for v in vals:
        ts = res[v]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            ts.step()
        torch.cuda.synchronize()
        print(f"{sys.argv[1]}({v}): C3 {(time.perf_counter() - t0) / 30 * 1e3:.4f} ms   loss {float(ts._loss):.6e}", flush=True)
