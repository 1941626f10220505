#!/usr/bin/env python3
"""Secondary measurements for BASELINE configs C3 / C4 / C5 -- parity-test shapes, not bench lines.
usage: python tools/bench_configs.py [c3] [c4] [c5]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
import numpy as np
import torch
from gaot_amd.model.gaot import GAOT
from gaot_amd.model.layers.magno import MAGNOConfig
from gaot_amd.model.layers.attn import TransformerConfig, AttentionConfig
from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch
from gaot_amd.trainer import TrainStep
from tests._workloads import naca_points, shell_points

dev = torch.device("cuda:0")


def grid(sizes):
    axes = [torch.linspace(-1, 1, n) for n in sizes]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), -1).reshape(-1, len(sizes)).to(dev)


def timed(fn, warm, iters):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def c3():
    torch.manual_seed(0)
    B, N = 16, 8192
    mc = MAGNOConfig(radius=0.033, lifting_channels=64, precompute_edges=True)
    model = GAOT(3, 1, NS(args=NS(magno=mc, transformer=TransformerConfig(patch_size=2, hidden_size=256)), latent_tokens_size=[64, 64])).to(dev).train()
    g = torch.Generator().manual_seed(0)
    lat = grid([64, 64])
    x = torch.stack([naca_points(N, g) for _ in range(B)]).to(dev)
    ns = NeighborSearch("native")
    enc = [[ns(x[b], lat, 0.033)] for b in range(B)]
    dec = [[ns(lat, x[b], 0.033)] for b in range(B)]
    deg = torch.cat([e[0]["neighbors_row_splits"][1:] - e[0]["neighbors_row_splits"][:-1] for e in enc])
    p, t = torch.randn(B, N, 3, device=dev), torch.randn(B, N, 1, device=dev)
    ts = TrainStep(model)
    ts.bind(p, t, latent_tokens_coord=lat, xcoord=x, encoder_nbrs=enc, decoder_nbrs=dec)
    dt = timed(ts.step, 5, 20)
    return {"config": "C3 NACA0012-like vx, 8192 nodes, batch 16", "train_samples_per_s": B / dt, "ms_per_step": dt * 1e3,
            "edges_per_sample": int(sum(e[0]["neighbors_index"].numel() for e in enc) / B),
            "encoder_degree_max": int(deg.max()), "encoder_degree_mean": float(deg.float().mean())}


def c4():
    torch.manual_seed(0)
    B, N = 4, 16384
    mc = MAGNOConfig(radius=0.033, lifting_channels=64)
    model = GAOT(4, 2, NS(args=NS(magno=mc, transformer=TransformerConfig(patch_size=2, hidden_size=256)), latent_tokens_size=[64, 64])).to(dev)
    lat = grid([64, 64])
    x = (torch.rand(N, 2, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
    xb = torch.randn(B, N, 4, device=dev)
    tgt = torch.randn(B, N, 2, device=dev)
    model.train()
    ts = TrainStep(model)
    ts.bind(xb, tgt, latent_tokens_coord=lat, xcoord=x)
    dt_train = timed(ts.step, 5, 20)
    model.eval()
    stats = {"u": {"mean": torch.zeros(2), "std": torch.ones(2)}, "der": {"mean": torch.zeros(2), "std": torch.ones(2)},
             "start_time": {"mean": 0.0, "std": 1.0}, "time_diffs": {"mean": 0.0, "std": 1.0}}
    tv, ti = np.linspace(0, 1, 21), np.arange(0, 22, 2)[:11]
    roll = lambda: model.autoregressive_predict(x_batch=xb[..., :2], time_indices=ti, t_values=tv, stats=stats, stepper_mode="time_der",
                                                latent_tokens_coord=lat, fixed_coord=x)
    dt_roll = timed(roll, 2, 5)
    return {"config": "C4 NS-Gauss-like fx, 16384 nodes, batch 4", "pair_train_samples_per_s": B / dt_train, "train_ms_per_step": dt_train * 1e3,
            "rollout_10_steps_ms": dt_roll * 1e3, "rollout_ms_per_step": dt_roll * 1e3 / 10}


def c5(build_only: bool = False):
    torch.manual_seed(0)
    B, N = 1, 65536
    mc = MAGNOConfig(coord_dim=3, radius=0.067, lifting_channels=48)
    model = GAOT(3, 1, NS(args=NS(magno=mc, transformer=TransformerConfig(patch_size=2, hidden_size=384, attn_config=AttentionConfig(num_heads=8, num_kv_heads=8))),
                          latent_tokens_size=[32, 32, 32])).to(dev).train()
    lat = grid([32, 32, 32])
    x = shell_points(N, torch.Generator().manual_seed(0)).to(dev)
    p, t = torch.randn(B, N, 3, device=dev), torch.randn(B, N, 1, device=dev)
    ts = TrainStep(model)
    ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
    if build_only:
        return ts
    dt = timed(ts.step, 3, 10)
    nb = list(model.encoder.neighbor_cache.values())[0][0]
    deg = nb["neighbors_row_splits"][1:] - nb["neighbors_row_splits"][:-1]
    return {"config": "C5 3-D point cloud 65536 nodes, latent 32^3, batch 1", "train_samples_per_s": B / dt, "ms_per_step": dt * 1e3,
            "encoder_edges": int(nb["neighbors_index"].numel()), "encoder_degree_max": int(deg.max()), "empty_latent_tokens": int((deg == 0).sum())}


if __name__ == "__main__":
    which = [a for a in sys.argv[1:]] or ["c3", "c4", "c5"]
    for w in which:
        print(json.dumps({"c3": c3, "c4": c4, "c5": c5}[w]()), flush=True)
