#!/usr/bin/env python3
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
for (M, N, K, tile) in [(4096, 4096, 4096, 1), (4096, 4096, 4096, 5), (8192, 2048, 256, 2), (8192, 2048, 256, 3), (8192, 256, 1024, 3), (8192, 256, 256, 3)]:
    A, B, out = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
    lib.gaot_debug_set_gemm_tile(tile)
    row = []
    for ab in (0, 1, 2, 3, 4, 7):
        lib.gaot_debug_set_gemm_ablate(ab)
        us = timeit(lambda: ops.gemm(M, N, K, A, K, 1, B, K, 1, out, N))
        row.append(f"abl{ab}: {us:7.1f}us {2.0*M*N*K/us/1e6:6.1f}TF")
    lib.gaot_debug_set_gemm_ablate(0); lib.gaot_debug_set_gemm_tile(0)
    print(f"M={M} N={N} K={K} tile={tile} | " + " | ".join(row), flush=True)
