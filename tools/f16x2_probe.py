#!/usr/bin/env python3
"""The two ways the "f32" precision can form its products on the split tiles -- fp16x2 (two fp16 pieces of the scaled operand, three
piece products; the default) and bf16x3 (three bf16 pieces, six piece products) -- next to the bf16x2 variant: time and error vs
float64 on the model's GEMM shapes at several operand magnitudes, incl. rows of very different magnitude inside one operand."""
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
rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
M = 8192
shapes = [("nt", 2048, 256), ("nt", 256, 1024), ("nt", 768, 256), ("nn", 256, 2048), ("nn", 1024, 256), ("nn", 256, 768), ("tn", 2048, 256), ("tn", 256, 1024)]
g = torch.Generator().manual_seed(0)
modes = [("fp16x2", lambda: (ops.set_precision("f32"), ops.set_f32_pieces("fp16x2"))), ("bf16x3", lambda: (ops.set_precision("f32"), ops.set_f32_pieces("bf16x3"))),
         ("bf16x2", lambda: ops.set_precision("bf16x2"))]
for mag_a, mag_b, rows in ((1.0, 0.06, None), (1e-6, 0.06, None), (300.0, 1e-3, None), (1.0, 0.06, 1e-4), (1.0, 0.06, 1e-7)):
    print(f"operand magnitudes A ~ {mag_a}, B ~ {mag_b}" + (f"; every other row of A scaled by {rows}" if rows else ""))
    for kind, N, K in shapes:
        if kind == "nt":
            A, B = torch.randn(M, K, generator=g) * mag_a, torch.randn(N, K, generator=g) * mag_b
        elif kind == "nn":
            A, B = torch.randn(M, K, generator=g) * mag_a, torch.randn(K, N, generator=g) * mag_b
        else:
            A, B = torch.randn(M, N, generator=g) * mag_a, torch.randn(M, K, generator=g) * mag_b
        if rows:
            A[1::2] *= rows
        ref = A.double() @ B.double().t() if kind == "nt" else (A.double() @ B.double() if kind == "nn" else A.double().t() @ B.double())
        Ad, Bd = A.to(dev), B.to(dev)
        out = torch.empty(N, K, device=dev) if kind == "tn" else None
        def f():
            if kind == "nt": return ops.linear_nt(Ad, Bd)
            if kind == "nn": return ops.matmul_nn(Ad, Bd)
            ops.wgrad_launch([(Ad, N, Bd, K, out, K, None, N, K, M)])
            return out
        row = []
        for name, setm in modes:
            setm()
            y = f()
            e = rel(y, ref)
            extra = ""
            if rows and kind != "tn":
                extra = f" small rows {rel(y[1::2], ref[1::2]):.1e}"
            t = timeit(f)
            row.append(f"{name}: {t:6.1f}us {e:.2e}{extra}")
        ops.set_precision("f32"); ops.set_f32_pieces("fp16x2")
        print(f"  {kind} N={N:5d} K={K:5d} | " + " | ".join(row), flush=True)
