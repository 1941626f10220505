#!/usr/bin/env python3
"""split-bf16 tile kernels: one workgroup per tile vs persistent workgroups (gaot_debug_set_split_persist), 128- and 64-row tiles,
on the transformer's shapes.  usage: python tools/gemm_persist_sweep.py"""
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
shapes = [("nt", 8192, 2048, 256, 1), ("nt", 8192, 768, 256, 1), ("nt", 8192, 256, 1024, 1), ("nt", 8192, 256, 256, 1),
          ("nn", 8192, 1024, 256, 1), ("nn", 8192, 256, 2048, 2), ("nn", 8192, 256, 768, 1),
          ("tn", 2048, 256, 8192, 16), ("tn", 256, 1024, 8192, 32), ("tn", 768, 256, 8192, 32), ("nt", 4096, 4096, 4096, 1)]
for kind, M, N, K, sk in shapes:
    out = torch.empty(M, N, device=dev)
    if kind == "nt":
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev); f = lambda: ops.gemm(M, N, K, A, K, 1, B, K, 1, out, N, split_k=sk)
    elif kind == "nn":
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev); f = lambda: ops.gemm(M, N, K, A, K, 1, B, N, 0, out, N, split_k=sk)
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev); f = lambda: ops.gemm(M, N, K, A, M, 0, B, N, 0, out, N, split_k=sk)
    row = []
    for mode, name in ((5, "bm128"), (7, "bm64")):
        lib.gaot_debug_set_gemm_glds(mode)
        for persist in (0, 512, 384, 256):
            lib.gaot_debug_set_split_persist(persist)
            us = timeit(f)
            row.append(f"{name} p{persist} {us:5.1f}")
    lib.gaot_debug_set_split_persist(0); lib.gaot_debug_set_gemm_glds(4)
    print(f"{kind} {M}x{N}x{K} sk{sk} | " + " | ".join(row), flush=True)
