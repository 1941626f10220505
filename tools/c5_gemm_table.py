#!/usr/bin/env python3
"""Per-launch GEMM table of one eager training step at the C5 shape (65 536-point 3-D cloud, 32^3 latent grid, 4 096 tokens x 384):
HIP events around every gaot_gemm_f32 call, with the path the dispatcher took (1 fp32 MFMA, 2 skinny, 3 split-bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib
import tools.bench_configs as BC
lib = _lib.load()
records = []
raw = ops.gemm
def timed_gemm(M, N, K, *a, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); out = raw(M, N, K, *a, **kw); e.record()
    records.append((s, e, 2.0 * M * N * K, (M, N, K, int(a[2]), int(a[5]), kw.get("split_k", 1)), lib.gaot_debug_last_gemm_path()))
    return out
ts = BC.c5(build_only=True)
ts.use_graph = False
ts.step(); torch.cuda.synchronize()
ops.gemm = timed_gemm
torch.cuda._sleep(int(60e-3 * 2.0e9))
ts.step(); torch.cuda.synchronize()
ops.gemm = raw
tot = 0.0
for s, e, fl, shp, path in sorted(records, key=lambda r: -r[0].elapsed_time(r[1])):
    us = s.elapsed_time(e) * 1e3; tot += us
    print(f"gemm M={shp[0]:6d} N={shp[1]:5d} K={shp[2]:6d} a_k={shp[3]} b_k={shp[4]} split={shp[5]:3d} path {path} {us:8.1f}us {fl / us / 1e6:6.1f}TF")
print(f"total {tot:.0f} us over {len(records)} launches")
