#!/usr/bin/env python3
"""same-process A/B of the fp16-piece attention backward variants (gaot_debug_set_attention_h16): rounds of 200 calls per mode, alternating;
minimum and median per mode.  usage: attn_h16_ab.py [B] mode mode ...   (0 = attn_bwd_split8_kernel, 8 / 0x18 / 0x28 / 0x38 = h16<8> variants, 4 = h16<4>)"""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib as L
dev = torch.device("cuda:0")
args = [int(a, 0) for a in sys.argv[1:]]
B, modes = args[0], args[1:]
torch.manual_seed(0)
qkv = torch.randn(B, 1024, 768, device=dev, requires_grad=True)
go = torch.randn(B, 1024, 256, device=dev)
ops.begin_pass()
o = ops.attention(qkv, 8, 8, 32)
def bwd(): torch.autograd.grad(o, qkv, go, retain_graph=True)
times = {m: [] for m in modes}
for rnd in range(7):
    for m in modes:
        L.load().gaot_debug_set_attention_h16(m)
        for _ in range(10): bwd()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(200): bwd()
        e.record(); torch.cuda.synchronize()
        times[m].append(s.elapsed_time(e) / 200 * 1e3)
for m in modes:
    print(f"B={B} mode {m:#x}: min {min(times[m]):.1f} us  median {statistics.median(times[m]):.1f} us (backward incl. dQ reduce)", flush=True)
