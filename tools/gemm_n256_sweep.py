#!/usr/bin/env python3
"""N = 256 products of the transformer (M = 8192 tokens): fp32-MFMA tiles vs split-bf16 tiles with split-K, per shape.
usage: python tools/gemm_n256_sweep.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib as L
dev = torch.device("cuda:0"); lib = L.load()
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
M, N = 8192, 256
for kind in ("nt", "nn"):
    for K in (256, 512, 768, 1024, 2048):
        A = torch.randn(M, K, device=dev)
        B = torch.randn(N, K, device=dev) if kind == "nt" else torch.randn(K, N, device=dev)
        out = torch.empty(M, N, device=dev)
        def f(sk):
            ws_split = sk
            if kind == "nt": ops.gemm(M, N, K, A, K, 1, B, K, 1, out, N, split_k=sk)
            else: ops.gemm(M, N, K, A, K, 1, B, N, 0, out, N, split_k=sk)
        row = []
        for mode, name in ((1, "fp32"), (5, "split128"), (7, "split64")):
            lib.gaot_debug_set_gemm_glds(mode)
            for sk in (1, 2, 4):
                if K // sk < 128: continue
                us = timeit(lambda: f(sk))
                row.append(f"{name} sk{sk} {us:5.1f}us({lib.gaot_debug_last_gemm_path()})")
        lib.gaot_debug_set_gemm_glds(4)
        us = timeit(lambda: f(1)); row.append(f"DEFAULT {us:5.1f}us({lib.gaot_debug_last_gemm_path()})")
        print(f"{kind} K={K:5d} | " + " | ".join(row), flush=True)
