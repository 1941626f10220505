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
def rel(a, b): return float((a.double().cpu() - b).norm() / b.norm())
g = torch.Generator().manual_seed(0)
for (M, N, K) in [(200, 132, 64), (520, 260, 96), (1000, 64, 256)]:
    x, w, gy = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(M, N, generator=g)
    for tile in (1, 2, 3):
        lib.gaot_debug_set_gemm_glds(1); lib.gaot_debug_set_gemm_tile(tile)
        y = ops.linear_nt(x.to(dev), w.to(dev), bias=torch.ones(N, device=dev))
        dx = ops.matmul_nn(gy.to(dev), w.to(dev))
        Mk = M - M % 32
        dw = torch.empty(N, K, device=dev); ops.gemm(N, K, Mk, gy.to(dev), N, 0, x.to(dev), K, 0, dw, K, split_k=2)
        lib.gaot_debug_set_gemm_glds(0); lib.gaot_debug_set_gemm_tile(0)
        e = (rel(y, x.double() @ w.double().t() + 1), rel(dx, gy.double() @ w.double()), rel(dw, gy[:Mk].double().t() @ x[:Mk].double()))
        print(f"M={M} N={N} K={K} tile={tile} rel err nt/nn/tn: {e[0]:.2e} {e[1]:.2e} {e[2]:.2e}", flush=True)
        assert max(e) < 3e-6
for (kind, M, N, K) in [("nt", 4096, 4096, 4096), ("nt", 8192, 2048, 256), ("nt", 8192, 768, 256), ("nt", 8192, 256, 1024), ("nt", 8192, 256, 256),
                        ("nn", 8192, 256, 2048), ("nn", 8192, 1024, 256), ("tn", 2048, 256, 8192), ("tn", 256, 1024, 8192), ("tn", 256, 256, 8192)]:
    out = torch.empty(M, N, device=dev)
    if kind == "nt":
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev); f = lambda: ops.gemm(M, N, K, A, K, 1, B, K, 1, out, N)
    elif kind == "nn":
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev); f = lambda: ops.gemm(M, N, K, A, K, 1, B, N, 0, out, N)
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev); sk = ops._split_for_reduction(M, N, K); f = lambda: ops.gemm(M, N, K, A, M, 0, B, N, 0, out, N, split_k=sk)
    row = []
    lib.gaot_debug_set_gemm_glds(0); us0 = timeit(f); row.append(f"reg {us0:7.1f}us {2.0*M*N*K/us0/1e6:6.1f}TF")
    for st in (3, 1):
      for tile in (0, 1, 2, 3):
        lib.gaot_debug_set_gemm_glds(st); lib.gaot_debug_set_gemm_tile(tile); us = timeit(f)
        row.append(f"ring{3 if st==3 else 2}t{tile} {us:6.1f}us {2.0*M*N*K/us/1e6:5.1f}")
    lib.gaot_debug_set_gemm_glds(1); lib.gaot_debug_set_gemm_tile(0)
    print(f"{kind} M={M} N={N} K={K} | " + " | ".join(row), flush=True)
