#!/usr/bin/env python3
"""Fused kernel-MLP kernels: time forward / backward at the bench shape and ablate the forward."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib as L
lib = L.load(); d = "cuda"
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(20e-3 * 2.0e9))       # park the GPU: the host enqueues the whole batch behind the spin
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
E, cin, n = 55592, 4, 4
x = torch.rand(E, cin, device=d) * 2 - 1
dims = [cin] + [64] * n
ws = [(torch.randn(dims[i + 1], dims[i], device=d) / dims[i] ** 0.5).requires_grad_() for i in range(n)]
bs = [(0.1 * torch.randn(64, device=d)).requires_grad_() for _ in range(n)]
acts = ["gelu"] * (n - 1) + ["none"]
dk = torch.randn(E, 64, device=d)
def fwd():
    with torch.no_grad(): return ops.mlp_chain(x, ws, bs, acts)
y = ops.mlp_chain(x, ws, bs, acts)
def bwd(): torch.autograd.grad(y, ws + bs, dk, retain_graph=True)
print(f"fwd {timeit(fwd):.1f} us   bwd (kernel + reduce) {timeit(bwd):.1f} us")
for bits, name in [(1, "no stores"), (2, "no gelu"), (4, "no mfma"), (8, "no staging"), (6, "no gelu+mfma"), (15, "nothing")]:
    lib.gaot_debug_set_kernel_mlp_ablate(bits); t = timeit(fwd); lib.gaot_debug_set_kernel_mlp_ablate(0)
    print(f"  fwd {name:14s} {t:.1f} us")
