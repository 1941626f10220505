#!/usr/bin/env python3
"""N = 256 products of the processor (M = 8192 rows): only 128-256 output tiles, i.e. <= 1 workgroup per CU.
Does split-K (more workgroups, plus a reduce pass) pay on the forward / input-gradient shapes?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib as L
dev = torch.device("cuda:0"); lib = L.load()
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (kind, M, N, K) in [("nn", 8192, 256, 2048), ("nt", 8192, 256, 1024), ("nn", 8192, 256, 768), ("nt", 8192, 256, 512), ("nt", 8192, 256, 256), ("nn", 8192, 256, 256),
                        ("nt", 8192, 768, 256), ("nn", 8192, 1024, 256)]:
    out = torch.empty(M, N, device=dev); res = torch.randn(M, N, device=dev)
    if kind == "nt":
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev); f = lambda sk: ops.gemm(M, N, K, A, K, 1, B, K, 1, out, N, split_k=sk, residual=res, ldr=N)
    else:
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev); f = lambda sk: ops.gemm(M, N, K, A, K, 1, B, N, 0, out, N, split_k=sk, residual=res, ldr=N)
    row = []
    for tile in (0, 1, 2, 3):
        lib.gaot_debug_set_gemm_tile(tile)
        for sk in (1, 2, 4):
            if K // sk < 128: continue
            us = timeit(lambda: f(sk)); row.append(f"t{tile}sk{sk} {us:5.1f}")
    lib.gaot_debug_set_gemm_tile(0)
    print(f"{kind} M={M} N={N} K={K} | " + " | ".join(row), flush=True)
