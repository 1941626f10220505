#!/usr/bin/env python3
"""Accumulation bias of the split-bf16 tile kernels: random-sign vs all-positive operands against float64, fp32-MFMA tiles beside them."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from gaot_amd import ops, _lib as L
lib = L.load(); dev = torch.device("cuda:0")
rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
torch.manual_seed(0)
for (M, N, K, kind) in ((8192, 256, 2048, "nn"), (8192, 2048, 256, "nt"), (2048, 256, 8192, "tn"), (4096, 4096, 4096, "nt")):
    for name, gen in (("random sign", lambda *s: torch.randn(*s)), ("all positive", lambda *s: torch.rand(*s) + 0.5)):
        if kind == "nt":
            A, B = gen(M, K), gen(N, K); ref = A.double() @ B.double().t()
            f = lambda: ops.linear_nt(A.to(dev), B.to(dev))
        elif kind == "nn":
            A, B = gen(M, K), gen(K, N); ref = A.double() @ B.double()
            f = lambda: ops.matmul_nn(A.to(dev), B.to(dev))
        else:
            A, B = gen(K, M), gen(K, N); ref = A.double().t() @ B.double()
            f = lambda: ops.matmul_tn(A.to(dev), B.to(dev))
        row = []
        for mode, nm in ((1, "fp32 MFMA"), (4, "default (split-bf16 where it applies)")):
            lib.gaot_debug_set_gemm_glds(mode)
            out = f(); row.append(f"{nm}: {rel(out, ref):.2e} (path {lib.gaot_debug_last_gemm_path()})")
        lib.gaot_debug_set_gemm_glds(4)
        print(f"{kind} {M}x{N}x{K} {name:12s} | " + " | ".join(row))
