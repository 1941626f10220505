#!/usr/bin/env python3
"""Forward / backward time of the processor's attention at the bench shape (8 x 1 024 tokens, 8 heads of 32) and at the 4 096-token one."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops
from gaot_amd import _lib as L
dev = torch.device("cuda:0")
if "--h16" in sys.argv:          # 4 / 8: the fp16-piece backward as attn_bwd_h16_kernel<4 / 8>
    L.load().gaot_debug_set_attention_h16(int(sys.argv[sys.argv.index("--h16") + 1]))
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for B in (8, 4):
    torch.manual_seed(0)
    qkv = torch.randn(B, 1024, 768, device=dev, requires_grad=True)
    go = torch.randn(B, 1024, 256, device=dev)
    ops.begin_pass()
    o = ops.attention(qkv, 8, 8, 32)
    def fwd():
        with torch.no_grad(): ops.attention(qkv, 8, 8, 32)
    def bwd(): torch.autograd.grad(o, qkv, go, retain_graph=True)
    g, = torch.autograd.grad(o, qkv, go, retain_graph=True)
    q, k, v = [t.double().reshape(B, 1024, 8, 32).transpose(1, 2) for t in qkv.detach().split(256, dim=-1)]
    qd, kd, vd = [t.clone().requires_grad_() for t in (q, k, v)]
    od = torch.softmax(qd @ kd.transpose(-1, -2) / 32 ** 0.5, -1) @ vd
    gq, gk, gv = torch.autograd.grad(od, (qd, kd, vd), go.double().reshape(B, 1024, 8, 32).transpose(1, 2))
    gref = torch.cat([t.transpose(1, 2).reshape(B, 1024, 256) for t in (gq, gk, gv)], -1)
    rel = lambda a, b: float((a.double() - b).norm() / b.norm())
    print(f"B={B}: fwd {timeit(fwd):.1f} us  bwd (incl. dq reduce) {timeit(bwd):.1f} us  out err {rel(o, od.transpose(1, 2).reshape(B, 1024, 256)):.2e}  grad err {rel(g, gref):.2e}", flush=True)
