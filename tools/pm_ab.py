#!/usr/bin/env python3
"""same-box A/B of the latent grid in patch-major order (GAOT._PATCH_MAJOR: patchify / unpatchify as reshapes) on C2, the 4 096-token batch
(C4-shaped) and C5: ms per TrainStep step, alternating; plus the dispatch count of one captured C2 step under each setting"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops
import tools.bench_configs as BC
import bench
from gaot_amd.trainer import TrainStep
from gaot_amd.model.gaot import GAOT
dev = torch.device("cuda:0")
res = {}
for v in (False, True):
    GAOT._PATCH_MAJOR[0] = v
    ops._PATH_CACHE.clear(); ops.register_grad_slots([], [])
    ts5 = BC.c5(build_only=True)
    out = []
    for nb in (4, 8):
        ops.register_grad_slots([], [])
        torch.manual_seed(0)
        m = bench.build_model().to(dev).train()
        lat, x, p, t = bench.synthetic(1234, dev)
        ts = TrainStep(m, lr=8e-4, weight_decay=1e-5, use_graph=True)
        ts.bind(p[:nb].contiguous(), t[:nb].contiguous(), latent_tokens_coord=lat, xcoord=x)
        out.append(ts)
    for ts in [ts5] + out:
        for _ in range(6):
            ts.step()
    torch.cuda.synchronize()
    res[v] = [ts5] + out
GAOT._PATCH_MAJOR[0] = True
for rnd in range(4):
    for v in (False, True):
        o = []
        for ts, n in zip(res[v], (20, 60, 60)):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                ts.step()
            torch.cuda.synchronize()
            o.append((time.perf_counter() - t0) / n * 1e3)
        print(f"patch_major={int(v)}: C5 {o[0]:.4f} ms  C4 {o[1]:.4f} ms  C2 {o[2]:.4f} ms   losses {float(res[v][0]._loss):.6e} {float(res[v][1]._loss):.6e} {float(res[v][2]._loss):.6e}", flush=True)
