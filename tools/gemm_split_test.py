#!/usr/bin/env python3
"""split-bf16 GEMM (gemm_split.hip) vs the fp32-MFMA kernels: accuracy against float64 and speed.
   gaot_debug_set_gemm_glds(5) forces the split kernel wherever it is eligible, (1) is the fp32 LDS-direct default."""
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
def maxerr(a, b, scale): return float(((a.double().cpu() - b).abs() / scale).max())
g = torch.Generator().manual_seed(0)
for (M, N, K) in [(200, 132, 64), (520, 260, 96), (1000, 64, 256), (4096, 512, 1024)]:
    x, w, gy = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(M, N, generator=g)
    x = x * torch.exp(3 * torch.randn(M, 1, generator=g))        # rows of very different magnitude
    Mk = M - M % 32
    ref = (x.double() @ w.double().t() + 1, gy.double() @ w.double(), gy[:Mk].double().t() @ x[:Mk].double())
    sc = (x.double().abs() @ w.double().abs().t() + 1, gy.double().abs() @ w.double().abs(), gy[:Mk].double().abs().t() @ x[:Mk].double().abs())
    for mode in (1, 5, 7, 11):
        lib.gaot_debug_set_gemm_glds(mode)
        y = ops.linear_nt(x.to(dev), w.to(dev), bias=torch.ones(N, device=dev)); p1 = lib.gaot_debug_last_gemm_path()
        dx = ops.matmul_nn(gy.to(dev), w.to(dev)); p2 = lib.gaot_debug_last_gemm_path()
        dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
        ops.gemm(N, K, Mk, gy.to(dev), N, 0, x.to(dev), K, 0, dw, K, split_k=2, colsum=db); p3 = lib.gaot_debug_last_gemm_path()
        e = (rel(y, ref[0]), rel(dx, ref[1]), rel(dw, ref[2]), rel(db, gy[:Mk].double().sum(0)))
        me = (maxerr(y, ref[0], sc[0]), maxerr(dx, ref[1], sc[1]), maxerr(dw, ref[2], sc[2]))
        print(f"M={M} N={N} K={K} mode={mode} paths={p1}{p2}{p3} rel nt/nn/tn/colsum: {e[0]:.2e} {e[1]:.2e} {e[2]:.2e} {e[3]:.2e}  max|err|/sum|a||b|: {me[0]:.2e} {me[1]:.2e} {me[2]:.2e}", flush=True)
        assert max(e) < 3e-6
lib.gaot_debug_set_gemm_glds(1)
for (kind, M, N, K) in [("nt", 4096, 4096, 4096), ("nt", 8192, 2048, 256), ("nt", 8192, 768, 256), ("nt", 8192, 256, 1024), ("nt", 8192, 256, 256),
                        ("nt", 8192, 256, 512), ("nn", 8192, 256, 2048), ("nn", 8192, 1024, 256), ("nn", 8192, 256, 768), ("nn", 8192, 256, 256),
                        ("tn", 2048, 256, 8192), ("tn", 256, 1024, 8192), ("tn", 768, 256, 8192), ("tn", 256, 256, 8192),
                        ("nt", 55592, 64, 64), ("nn", 55592, 64, 64), ("tn", 64, 64, 55584)]:
    out = torch.empty(M, N, device=dev)
    if kind == "nt":
        A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev); f = lambda sk=1: ops.gemm(M, N, K, A, K, 1, B, K, 1, out, N)
    elif kind == "nn":
        A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev); f = lambda sk=1: ops.gemm(M, N, K, A, K, 1, B, N, 0, out, N)
    else:
        A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev); f = lambda sk=1: ops.gemm(M, N, K, A, M, 0, B, N, 0, out, N, split_k=sk)
    row = []
    sk0 = ops._split_for_reduction(M, N, K) if kind == "tn" else 1
    lib.gaot_debug_set_gemm_glds(1); us0 = timeit(lambda: f(sk0)); row.append(f"fp32 sk{sk0} {us0:7.1f}us {2.0*M*N*K/us0/1e6:6.1f}TF")
    lib.gaot_debug_set_gemm_glds(5)
    sks = [1] if kind != "tn" else sorted({max(1, sk0 // 4), max(1, sk0 // 2), sk0, sk0 * 2, sk0 * 4})
    for sk in sks:
        us = timeit(lambda: f(sk)); row.append(f"split sk{sk} {us:6.1f}us {2.0*M*N*K/us/1e6:6.1f}TF")
    lib.gaot_debug_set_gemm_glds(11)
    for sk in ([1] if kind != "tn" else [sk0, sk0 * 2]):
        us = timeit(lambda: f(sk)); row.append(f"split256 sk{sk} {us:6.1f}us {2.0*M*N*K/us/1e6:6.1f}TF")
    lib.gaot_debug_set_gemm_glds(1)
    print(f"{kind} M={M} N={N} K={K} | " + " | ".join(row), flush=True)
