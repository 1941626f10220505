#!/usr/bin/env python3
"""GPU cost of (re)planning the block-diagonal union of a vx batch on skewed meshes (C3 shape: 16 samples x 8192 nodes):
union planned afresh vs composed from cached per-sample plans, and one training step with a SHUFFLED batch order per step.
usage: python tools/vx_replan_cost.py"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
import torch
from gaot_amd.plan import MergedGeometry
from gaot_amd.model.gaot import GAOT
from gaot_amd.model.layers.magno import MAGNOConfig
from gaot_amd.model.layers.attn import TransformerConfig
from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch
from gaot_amd.trainer import TrainStep
from tests._workloads import grid, naca_points

dev = torch.device("cuda:0")
B, N, POOL = 16, 8192, 48
g = torch.Generator().manual_seed(0)
lat = grid([64, 64]).to(dev)
xs = [naca_points(N, g, 0.15).to(dev) for _ in range(POOL)]
ns = NeighborSearch("native")
torch.cuda.synchronize(); t0 = time.perf_counter()
enc = [ns(xs[b], lat, 0.033) for b in range(POOL)]
dec = [ns(lat, xs[b], 0.033) for b in range(POOL)]
torch.cuda.synchronize()
out = {"radius_search_ms_per_sample_pair": 1e3 * (time.perf_counter() - t0) / POOL}


def gpu_ms(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters, 1e3 * (time.perf_counter() - t0) / iters


sel = list(range(B))
for name, dicts, src, dst in (("encoder", enc, lambda i: xs[i], lambda i: lat), ("decoder", dec, lambda i: lat, lambda i: xs[i])):
    mk = lambda: MergedGeometry([dicts[i] for i in sel], [src(i) for i in sel], [dst(i) for i in sel])
    out[f"{name}_union_planned_afresh_gpu_ms,wall_ms"] = gpu_ms(mk)
    MergedGeometry([dicts[i] for i in sel], [src(i) for i in sel], [dst(i) for i in sel], build_parts=True)
    assert mk().composed
    out[f"{name}_union_composed_gpu_ms,wall_ms"] = gpu_ms(mk)
    full = lambda: (lambda m: (m.geo_stats(), m.plan.edge_features(m.src, m.dst), m.plan.cosine_attention(m.src, m.dst)))(mk())
    out[f"{name}_union_composed_plus_statistics_features_attention_gpu_ms,wall_ms"] = gpu_ms(full)

# training with a shuffled batch every step (eager: the union changes per step)
torch.manual_seed(0)
mc = MAGNOConfig(radius=0.033, lifting_channels=64, precompute_edges=True)
model = GAOT(3, 1, NS(args=NS(magno=mc, transformer=TransformerConfig(patch_size=2, hidden_size=256)), latent_tokens_size=[64, 64])).to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=8e-4, weight_decay=1e-5, fused=True)
x_all = torch.stack(xs)
p_all, t_all = torch.randn(POOL, N, 3, device=dev), torch.randn(POOL, N, 1, device=dev)
from gaot_amd import ops


def step(idx):
    opt.zero_grad()
    pred = model(latent_tokens_coord=lat, xcoord=x_all[idx], pndata=p_all[idx], encoder_nbrs=[[enc[i]] for i in idx], decoder_nbrs=[[dec[i]] for i in idx])
    ops.mse_loss(pred, t_all[idx]).backward()
    opt.step()


gen = torch.Generator().manual_seed(1)
for phase in ("first epoch (unions planned afresh)", "second epoch (per-sample plans get built)", "third epoch (unions composed)"):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 0
    for _ in range(2):
        perm = torch.randperm(POOL, generator=gen).tolist()
        for s in range(0, POOL, B):
            step(perm[s:s + B]); n += 1
    torch.cuda.synchronize()
    out[f"shuffled eager training, {phase}: ms/step"] = 1e3 * (time.perf_counter() - t0) / n
ts = TrainStep(model)
idx = list(range(B))
ts.bind(p_all[idx], t_all[idx], latent_tokens_coord=lat, xcoord=x_all[idx], encoder_nbrs=[[enc[i]] for i in idx], decoder_nbrs=[[dec[i]] for i in idx])
for _ in range(5):
    ts.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    ts.step()
torch.cuda.synchronize()
out["fixed batch, hipGraph TrainStep: ms/step"] = 1e3 * (time.perf_counter() - t0) / 20
print(json.dumps(out, indent=1))
