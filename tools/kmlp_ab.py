#!/usr/bin/env python3
"""Kernel-MLP kernels A/B: bf16-split (default) against the fp32-MFMA kernels at the bench shapes -- time and error vs float64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib as L
lib = L.load(); d = "cuda"
def timeit(fn, iters=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(20e-3 * 2.0e9))
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for E, cin, n, act in [(55592, 4, 4, "gelu"), (55592, 7, 3, "relu"), (400000, 4, 4, "gelu")]:
    torch.manual_seed(0)
    x = torch.rand(E, cin, device=d) * 2 - 1
    dims = [cin] + [64] * n
    ws = [(torch.randn(dims[i + 1], dims[i], device=d) / dims[i] ** 0.5).requires_grad_() for i in range(n)]
    bs = [(0.1 * torch.randn(64, device=d)).requires_grad_() for _ in range(n)]
    acts = [act] * (n - 1) + ["none"]
    dk = torch.randn(E, 64, device=d)
    h = x.double()
    wd = [w.detach().double().requires_grad_() for w in ws]; bd = [b.detach().double().requires_grad_() for b in bs]
    for i in range(n):
        h = h @ wd[i].t() + bd[i]
        if i < n - 1: h = torch.nn.functional.gelu(h) if act == "gelu" else torch.relu(h)
    gd = torch.autograd.grad(h, wd + bd, dk.double())
    for split in (0, 1, 2):
        lib.gaot_debug_set_kernel_mlp_split(1 if split else 0); ops.set_gemm_pieces(2 if split == 2 else 3)
        def fwd():
            with torch.no_grad(): return ops.mlp_chain(x, ws, bs, acts)
        y = ops.mlp_chain(x, ws, bs, acts)
        def bwd(): return torch.autograd.grad(y, ws + bs, dk, retain_graph=True)
        g = bwd()
        errs = [rel(a, b) for a, b in zip(g, gd)]
        print(f"E={E} cin={cin} layers={n} {act} split={split}: fwd {timeit(fwd):6.1f} us  bwd {timeit(bwd):6.1f} us   out err {rel(y, h):.2e}  "
              f"grad err max {max(errs):.2e} (dW {' '.join(f'{e:.1e}' for e in errs[:n])} | db {' '.join(f'{e:.1e}' for e in errs[n:])})", flush=True)
    lib.gaot_debug_set_kernel_mlp_split(1)
