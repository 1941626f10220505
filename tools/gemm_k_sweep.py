#!/usr/bin/env python3
"""Fixed cost of a tile GEMM launch: time against K at the model's output shapes (two-piece products); the intercept of the fit is what
prologue + epilogue + launch ramp cost, the slope the k-loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib
if len(sys.argv) > 1: _lib.load().gaot_debug_set_gemm_glds(int(sys.argv[1]))
dev = torch.device("cuda:0")
def timeit(fn, iters=40):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(10e-3 * 2.0e9)); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
M = 8192
for kind, N in (("nt", 2048), ("nt", 768), ("nn", 256), ("nn", 1024)):
    row = []
    for K in (32, 64, 128, 256, 512, 1024):
        out = torch.empty(M, N, device=dev)
        if kind == "nt":
            A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev); f = lambda: ops.linear_nt(A, B, out=out, split_k=1)
        else:
            A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev); f = lambda: ops.matmul_nn(A, B, out=out, split_k=1)
        row.append((K, timeit(f)))
    (k0, t0), (k1, t1) = row[2], row[4]
    slope = (t1 - t0) / (k1 - k0)
    print(f"{kind} M={M} N={N}: " + "  ".join(f"K={k}: {t:.1f}" for k, t in row) + f"   slope {slope * 256:.1f} us per 256 k, intercept {t0 - slope * k0:.1f} us", flush=True)
