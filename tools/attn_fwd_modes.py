#!/usr/bin/env python3
"""Forward attention time per shape with the 4-wave (split mode 3) and the 8-wave (mode 2) workgroups forced, and the default:
   python tools/attn_fwd_modes.py        (C5: 1 x 4096 x 8 x 48; C4: 4 x 1024 x 8 x 32; C2: 8 x 1024 x 8 x 32; C3: 16 x 1024 x 8 x 32)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops
from gaot_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
def timeit(fn, iters=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (B, S, H, D) in ((1, 4096, 8, 48), (4, 1024, 8, 32), (8, 1024, 8, 32), (16, 1024, 8, 32)):
    torch.manual_seed(0)
    qkv = torch.randn(B, S, 3 * H * D, device=dev)
    ops.begin_pass()
    def fwd():
        with torch.no_grad(): ops.attention(qkv, H, H, D)
    res = {}
    for mode in (1, 3, 2):
        old = lib.gaot_debug_set_attention_split(mode)
        res[mode] = timeit(fwd)
        lib.gaot_debug_set_attention_split(old)
    old = lib.gaot_debug_set_attention_keysplit(0)
    res[0] = timeit(fwd)
    lib.gaot_debug_set_attention_keysplit(old)
    print(f"B={B} S={S} H={H} D={D}: default {res[1]:.1f} us (key split off: {res[0]:.1f})   4-wave {res[3]:.1f} us   8-wave {res[2]:.1f} us", flush=True)
