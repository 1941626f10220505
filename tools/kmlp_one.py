#!/usr/bin/env python3
"""A few forward + backward launches of the kernel MLP at one shape (for rocprofv3 passes): kmlp_one.py gelu|relu layers cin E [split]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib as L
lib = L.load(); d = "cuda"
act, n, cin, E = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
if len(sys.argv) > 5: lib.gaot_debug_set_kernel_mlp_split(int(sys.argv[5]))
torch.manual_seed(0)
x = torch.rand(E, cin, device=d) * 2 - 1
dims = [cin] + [64] * n
ws = [(torch.randn(dims[i + 1], dims[i], device=d) / dims[i] ** 0.5).requires_grad_() for i in range(n)]
bs = [(0.1 * torch.randn(64, device=d)).requires_grad_() for _ in range(n)]
acts = [act] * (n - 1) + ["none"]
dk = torch.randn(E, 64, device=d)
for _ in range(6):
    y = ops.mlp_chain(x, ws, bs, acts)
    torch.autograd.grad(y, ws + bs, dk)
torch.cuda.synchronize()
