#!/usr/bin/env python3
"""Run ONE GEMM shape a few times (for rocprofv3 --pmc).  usage: one_gemm.py nt|nn|tn M N K [tile]
ONE_GEMM_WEIGHT=1: B is registered as a parameter first (magnitude word + pre-split fp16 planes, as in the model)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib as L
kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
tile = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dev = torch.device("cuda:0")
L.load().gaot_debug_set_gemm_tile(tile)
out = torch.empty(M, N, device=dev)
if kind == "nt":
    A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    f = lambda: ops.gemm(M, N, K, A, K, 1, B, K, 1, out, N)
elif kind == "nn":
    A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
    f = lambda: ops.gemm(M, N, K, A, K, 1, B, N, 0, out, N)
else:
    A, B = torch.randn(K, M, device=dev), torch.randn(K, N, device=dev)
    f = lambda: ops.gemm(M, N, K, A, M, 0, B, N, 0, out, N, split_k=ops._split_for_reduction(M, N, K))
if os.environ.get("ONE_GEMM_WEIGHT") == "1" and kind != "tn":
    prm = torch.nn.Parameter(B)
    ops.begin_pass()
    ops.refresh_weight_amax([prm])
for _ in range(10):
    f()
torch.cuda.synchronize()
