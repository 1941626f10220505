#!/usr/bin/env python3
"""GEMM micro-benchmark on the GPU box: the model's GEMM shapes x tile configs -> TFLOP/s (fp32 MFMA peak 157.3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib as L

dev = torch.device("cuda:0")
lib = L.load()
TILES = {1: "128x128", 2: "128x64", 3: "64x64", 4: "128x32", 5: "128x128w8", 6: "128x64w8"}

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us

def run(kind, M, N, K, tiles=(1, 2, 3), split=None):
    """kind nt: C[M,N]=A[M,K]W[N,K]^T ; nn: C[M,N]=A[M,K]B[K,N] ; tn: C[M,N]=A[K,M]^T B[K,N] (K = long reduction)"""
    if kind == "nt":
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
        call = lambda sk: ops.gemm(M, N, K, A, K, 1, B, K, 1, out, N, split_k=sk)
    elif kind == "nn":
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
        call = lambda sk: ops.gemm(M, N, K, A, K, 1, B, N, 0, out, N, split_k=sk)
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
        call = lambda sk: ops.gemm(M, N, K, A, M, 0, B, N, 0, out, N, split_k=sk)
    out = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    res = []
    splits = split if split else [1]
    for t in tiles:
        lib.gaot_debug_set_gemm_tile(t)
        for sk in splits:
            us = timeit(lambda: call(sk))
            res.append(f"{TILES[t]}{'/s'+str(sk) if sk>1 else ''}: {us:7.1f}us {fl/us/1e6:6.1f}TF")
    lib.gaot_debug_set_gemm_tile(0)
    us = timeit(lambda: call(splits[0] if kind != 'tn' else ops._split_for_reduction(M, N, K)))
    print(f"{kind} M={M:6d} N={N:5d} K={K:6d} {fl/1e9:7.2f}GF | " + " | ".join(res) + f" | auto: {us:7.1f}us {fl/us/1e6:6.1f}TF", flush=True)

import os
if os.environ.get("FOCUS"):
    run("nt", 4096, 4096, 4096, tiles=(1, 5))
    for (N, K) in [(256, 256), (768, 256), (2048, 256), (256, 1024)]:
        run("nt", 8192, N, K, tiles=(2, 3, 5, 6))
    for (N, K) in [(256, 2048), (1024, 256)]:
        run("nn", 8192, N, K, tiles=(2, 3, 5, 6))
    for (M, N) in [(2048, 256), (256, 1024)]:
        run("tn", M, N, 8192, tiles=(2, 3, 5, 6), split=[ops._split_for_reduction(M, N, 8192)])
    sys.exit(0)
print("== big squares (kernel ceiling)")
run("nt", 4096, 4096, 4096, tiles=(1,))
run("nt", 8192, 8192, 512, tiles=(1, 2))
print("== processor forward (NT)")
for (N, K) in [(256, 256), (768, 256), (2048, 256), (256, 1024), (256, 512)]:
    run("nt", 8192, N, K)
print("== processor input-grad (NN)")
for (N, K) in [(256, 256), (256, 768), (256, 2048), (1024, 256), (512, 256)]:
    run("nn", 8192, N, K)
print("== processor weight-grad (TN), K = 8192 rows")
for (M, N) in [(256, 256), (768, 256), (2048, 256), (256, 1024), (256, 512)]:
    run("tn", M, N, 8192, split=[1, 2, 4, 8, 16])
print("== MAGNO (C = 64)")
run("nt", 32768, 64, 64); run("nt", 131072, 64, 64); run("nt", 55638, 64, 64); run("nt", 55638, 64, 4, tiles=(3,))
run("nn", 131072, 64, 64); run("nn", 55638, 64, 64)
run("tn", 64, 64, 131072, split=[16, 64, 256]); run("tn", 64, 64, 55638, split=[16, 64, 256])
run("nt", 131072, 1, 64, tiles=(4,)); run("nt", 131072, 64, 1, tiles=(3,))
