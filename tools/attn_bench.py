#!/usr/bin/env python3
"""head_dim-32 attention at the bench shape: fp32-MFMA kernels (mode 0) vs split-bf16 (1: 4-wave, 2: 8-wave forward)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib as L
lib = L.load(); dev = torch.device("cuda:0")
qkv = torch.randn(8, 1024, 768, device=dev, requires_grad=True)
go = torch.randn(8, 1024, 256, device=dev)
def timeit(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(20e-3 * 2.0e9)); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
ref = None
for mode in (0, 1, 2):
    lib.gaot_debug_set_attention_split(mode)
    with torch.no_grad():
        o = ops.attention(qkv, 8, 8, 32)
        tf = timeit(lambda: ops.attention(qkv, 8, 8, 32))
    o2 = ops.attention(qkv, 8, 8, 32)
    tb = timeit(lambda: torch.autograd.grad(o2, qkv, go, retain_graph=True))
    if ref is None: ref = o
    print(f"mode {mode}: fwd {tf:.1f} us   bwd (delta + main + dq reduce) {tb:.1f} us   max |o - o_fp32| {float((o - ref).abs().max()):.2e}")
lib.gaot_debug_set_attention_split(1)
