#!/usr/bin/env python3
"""narrow-output activation-side products with a long reduction (du @ [w1;w3]: 8192 x 256 x 2048 NN; w2: 8192 x 256 x 1024 NT): split-K
factor x tile height, product + reduce launch.  usage: python tools/gemm_splitk_sweep.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib as L
lib = L.load(); dev = torch.device("cuda:0")
def timeit(fn, iters=30):
    for _ in range(4): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(10e-3 * 2.0e9)); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
M = 8192
for kind, N, K in (("nn", 256, 2048), ("nt", 256, 1024), ("nn", 256, 768), ("nn", 1024, 256)):
    A = torch.randn(M, K, device=dev)
    B = torch.randn(K, N, device=dev) if kind == "nn" else torch.randn(N, K, device=dev)
    ref = (A.double() @ (B.double() if kind == "nn" else B.double().t())).cpu()
    out = torch.empty(M, N, device=dev)
    print(f"{kind} M={M} N={N} K={K}")
    for mode, name in ((5, "128-row"), (7, "64-row")):
        lib.gaot_debug_set_gemm_glds(mode)
        row = []
        for sk in (1, 2, 4, 8):
            if K // sk < 256: continue
            f = (lambda: ops.matmul_nn(A, B, out=out, split_k=sk)) if kind == "nn" else (lambda: ops.linear_nt(A, B, out=out, split_k=sk))
            try:
                y = f(); path = lib.gaot_debug_last_gemm_path()
                e = float((y.double().cpu() - ref).norm() / ref.norm())
                row.append(f"sk{sk}: {timeit(f):6.1f}us({path}) {e:.1e}")
            except Exception as ex:
                row.append(f"sk{sk}: {type(ex).__name__}")
        print(f"   {name:8s} " + " | ".join(row), flush=True)
lib.gaot_debug_set_gemm_glds(4)
