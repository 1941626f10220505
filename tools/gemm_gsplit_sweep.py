#!/usr/bin/env python3
"""plane kernel (gemm_split.hip) vs LDS-direct register-split kernel (gemm_gsplit.hip) on the transformer's shapes; accuracy vs
float64 first.  usage: python tools/gemm_gsplit_sweep.py"""
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
def rel(a, b): return float((a.double().cpu() - b).norm() / b.norm())
g = torch.Generator().manual_seed(0)
for (M, N, K) in [(200, 132, 64), (520, 260, 96), (1000, 128, 256), (4096, 512, 1024)]:
    x, w, gy = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(M, N, generator=g)
    x = x * torch.exp(3 * torch.randn(M, 1, generator=g))
    Mk = M - M % 32
    ref = (x.double() @ w.double().t() + 1, gy.double() @ w.double(), gy[:Mk].double().t() @ x[:Mk].double())
    for gs in (0, 1):
        lib.gaot_debug_set_gemm_gsplit(gs)
        for mode in (5, 7):
            lib.gaot_debug_set_gemm_glds(mode)
            y = ops.linear_nt(x.to(dev), w.to(dev), bias=torch.ones(N, device=dev)); p1 = lib.gaot_debug_last_gemm_path()
            dx = ops.matmul_nn(gy.to(dev), w.to(dev), split_k=1); p2 = lib.gaot_debug_last_gemm_path()
            dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
            ops.gemm(N, K, Mk, gy.to(dev), N, 0, x.to(dev), K, 0, dw, K, split_k=2, colsum=db); p3 = lib.gaot_debug_last_gemm_path()
            e = (rel(y, ref[0]), rel(dx, ref[1]), rel(dw, ref[2]), rel(db, gy[:Mk].double().sum(0)))
            print(f"M={M} N={N} K={K} gsplit={gs} mode={mode} paths={p1}{p2}{p3} rel nt/nn/tn/colsum: {e[0]:.2e} {e[1]:.2e} {e[2]:.2e} {e[3]:.2e}", flush=True)
            assert max(e) < 3e-6
lib.gaot_debug_set_gemm_gsplit(0); lib.gaot_debug_set_gemm_glds(4)
shapes = [("nt", 8192, 2048, 256, 1), ("nt", 8192, 768, 256, 1), ("nt", 8192, 256, 1024, 1), ("nt", 8192, 256, 256, 1), ("nt", 8192, 256, 512, 1),
          ("nn", 8192, 1024, 256, 1), ("nn", 8192, 256, 2048, 2), ("nn", 8192, 256, 2048, 1), ("nn", 8192, 256, 768, 1), ("nn", 8192, 256, 256, 1),
          ("tn", 2048, 256, 8192, 16), ("tn", 256, 1024, 8192, 32), ("tn", 768, 256, 8192, 32), ("tn", 256, 256, 8192, 32), ("nt", 4096, 4096, 4096, 1)]
for kind, M, N, K, sk in shapes:
    out = torch.empty(M, N, device=dev)
    if kind == "nt":
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev); f = lambda: ops.gemm(M, N, K, A, K, 1, B, K, 1, out, N, split_k=sk)
    elif kind == "nn":
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev); f = lambda: ops.gemm(M, N, K, A, K, 1, B, N, 0, out, N, split_k=sk)
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev); f = lambda: ops.gemm(M, N, K, A, M, 0, B, N, 0, out, N, split_k=sk)
    row = []
    lib.gaot_debug_set_gemm_glds(1); row.append(f"fp32 {timeit(f):5.1f}")
    for gs, nm in ((0, "plane"), (1, "gsplit")):
        lib.gaot_debug_set_gemm_gsplit(gs)
        for mode, name in ((5, "128"), (7, "64")):
            lib.gaot_debug_set_gemm_glds(mode)
            us = timeit(f); row.append(f"{nm}{name} {us:5.1f}({lib.gaot_debug_last_gemm_path()})")
    lib.gaot_debug_set_gemm_gsplit(0); lib.gaot_debug_set_gemm_glds(4)
    row.append(f"DEFAULT {timeit(f):5.1f}")
    print(f"{kind} {M}x{N}x{K} sk{sk} | " + " | ".join(row) + f" | best possible {2.0 * M * N * K / 419e6:5.1f}us at 419TF", flush=True)
