#!/usr/bin/env python3
"""head_dim-32 attention at the bench shape: fp32-MFMA kernels (mode 0) vs split-bf16 (1: 4-wave, 2: 8-wave forward), and the
software-pipelined variants of the split kernels on / off; errors against float64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib as L
lib = L.load(); dev = torch.device("cuda:0")
B, S, H, D = int(os.environ.get("B", 8)), int(os.environ.get("S", 1024)), 8, int(os.environ.get("D", 32))
qkv = torch.randn(B, S, 3 * H * D, device=dev, requires_grad=True)
go = torch.randn(B, S, H * D, device=dev)
def timeit(fn, iters=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(20e-3 * 2.0e9)); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
q64 = qkv.detach().double().requires_grad_(True)
q_, k_, v_ = (t.reshape(B, S, H, D).transpose(1, 2) for t in q64.split(H * D, dim=2))
o64 = torch.softmax(q_ @ k_.transpose(-1, -2) / D ** 0.5, -1) @ v_
o64 = o64.transpose(1, 2).reshape(B, S, H * D)
g64, = torch.autograd.grad(o64, q64, go.double())
rel = lambda a, b: float((a.double() - b).norm() / b.norm())
for mode, pipe, pp, op in ((0, 0, 33, 3), (2, 0, 33, 3), (2, 0, 22, 3), (2, 0, 22, 2), (2, 0, 33, 3), (2, 0, 22, 3), (2, 0, 22, 2)):
    lib.gaot_debug_set_attention_split(mode)
    lib.gaot_debug_set_attention_pipe(pipe)
    lib.gaot_debug_set_attention_p_pieces(pp)
    lib.gaot_debug_set_attention_operand_pieces(op)
    with torch.no_grad():
        o = ops.attention(qkv, H, H, D)
        tf = timeit(lambda: ops.attention(qkv, H, H, D))
    o2 = ops.attention(qkv, H, H, D)
    g, = torch.autograd.grad(o2, qkv, go, retain_graph=True)
    tb = timeit(lambda: torch.autograd.grad(o2, qkv, go, retain_graph=True))
    print(f"mode {mode} pipe {pipe} P pieces {pp} operand pieces {op}: fwd {tf:.1f} us   bwd (delta + main + dq reduce) {tb:.1f} us   rel-L2 vs f64: out {rel(o, o64.detach()):.2e}  dqkv {rel(g, g64):.2e}")
lib.gaot_debug_set_attention_split(1); lib.gaot_debug_set_attention_pipe(0); lib.gaot_debug_set_attention_p_pieces(22); lib.gaot_debug_set_attention_operand_pieces(2)
