#!/usr/bin/env python3
"""same-box A/B of an integer debug switch of the library (default gaot_debug_set_gemm_ad_narrow; or: c45_ab.py gaot_debug_set_X v1 v2) on the 4 096-token configurations: C4 (N = 256 products) and C5 (N = 384): ms per TrainStep step"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib
lib = _lib.load()
import tools.bench_configs as BC
import bench
from gaot_amd.trainer import TrainStep
fname = "gaot_debug_set_gemm_ad_narrow"
argv = sys.argv[1:]
if argv and argv[0].startswith("gaot_"):
    fname, argv = argv[0], argv[1:]
vals = [int(v) for v in argv] or [1, 5]
setter = getattr(lib, fname)
dev = torch.device("cuda:0")
res = {}
for v in vals:
    old = setter(v)
    ops._PATH_CACHE.clear(); ops.register_grad_slots([], [])
    ts5 = BC.c5(build_only=True)
    ops.register_grad_slots([], [])
    torch.manual_seed(0)
    m4 = bench.build_model().to(dev).train()
    lat, x, p, t = bench.synthetic(1234, dev)
    ts4 = TrainStep(m4, lr=8e-4, weight_decay=1e-5, use_graph=True)
    ts4.bind(p[:4].contiguous(), t[:4].contiguous(), latent_tokens_coord=lat, xcoord=x)
    for ts in (ts5, ts4):
        for _ in range(6):
            ts.step()
    torch.cuda.synchronize()
    res[v] = (ts5, ts4)
    setter(old)
for rnd in range(3):
    for v in vals:
        out = []
        for ts, n in zip(res[v], (20, 60)):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                ts.step()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / n * 1e3)
        print(f"{fname}({v}): C5 {out[0]:.4f} ms  C4 {out[1]:.4f} ms   losses {float(res[v][0]._loss):.6e} {float(res[v][1]._loss):.6e}", flush=True)
