#!/usr/bin/env python3
"""attn_bwd_h16_kernel<8 / 4> against attn_bwd_split8_kernel (fp16 pieces): which of dQ / dK / dV are bit-identical, run-to-run determinism"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib as L
dev = torch.device("cuda:0")
for B, S in ((8, 1024), (16, 1024), (8, 2048), (8, 1000)):
    torch.manual_seed(0)
    qkv = torch.randn(B, S, 768, device=dev, requires_grad=True)
    go = torch.randn(B, S, 256, device=dev)
    res = {}
    for mode in (0, 8, 4, 8, 4):
        L.load().gaot_debug_set_attention_h16(mode)
        ops.begin_pass()
        o = ops.attention(qkv, 8, 8, 32)
        g, = torch.autograd.grad(o, qkv, go)
        res.setdefault(mode, []).append(g.clone())
    torch.cuda.synchronize()
    def thirds(a, b):
        return [bool(torch.equal(a[..., 256 * i:256 * (i + 1)], b[..., 256 * i:256 * (i + 1)])) for i in range(3)]
    print(B, S, "8 vs 0 (dQ, dK, dV):", thirds(res[8][0], res[0][0]), " 4 vs 0:", thirds(res[4][0], res[0][0]),
          " repeat 8:", torch.equal(res[8][0], res[8][1]), " repeat 4:", torch.equal(res[4][0], res[4][1]),
          " dQ 8 vs 0 max rel:", float((res[8][0][..., :256] - res[0][0][..., :256]).abs().max() / res[0][0][..., :256].abs().max()), flush=True)
