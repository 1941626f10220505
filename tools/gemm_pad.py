#!/usr/bin/env python3
"""Does a non-power-of-two leading dimension remove the in-loop load cost? (channel hot-spotting test)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib as L
dev = torch.device("cuda:0"); lib = L.load()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (M, N, K) in [(4096, 4096, 4096), (8192, 2048, 256), (8192, 256, 1024), (8192, 256, 2048), (8192, 768, 256), (8192, 256, 256)]:
    row = []
    for pad in (0, 32, 48, 8):
        A = torch.randn(M, K + pad, device=dev)[:, :K]
        B = torch.randn(N, K + pad, device=dev)[:, :K]
        out = torch.empty(M, N, device=dev)
        us = timeit(lambda: ops.gemm(M, N, K, A, K + pad, 1, B, K + pad, 1, out, N))
        row.append(f"pad{pad}: {us:7.1f}us {2.0*M*N*K/us/1e6:6.1f}TF")
    print(f"nt M={M} N={N} K={K} | " + " | ".join(row), flush=True)
