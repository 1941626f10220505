#!/usr/bin/env python3
"""The model's activation-side GEMM shapes (M = 8192 tokens) under the kernel-choice modes, with two-piece products: which tile
kernel wins per shape now?  modes: 4 = heuristic, 5 = split 128-row tiles wherever eligible, 7 = split 64-row tiles wherever eligible,
1 = fp32-MFMA tiles (gemm_glds) only."""
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
M = 8192
shapes = [("nt", 2048, 256), ("nt", 256, 1024), ("nt", 768, 256), ("nt", 256, 256), ("nt", 256, 512),
          ("nn", 256, 2048), ("nn", 1024, 256), ("nn", 256, 768), ("nn", 256, 256), ("nn", 256, 512), ("nn", 512, 256)]
if len(sys.argv) > 1:
    shapes = [s_ for s_ in shapes if s_[0] in sys.argv[1:]]
for pieces in ("fp16x2", "bf16x3", "bf16x2"):
    if pieces == "bf16x2": ops.set_precision("bf16x2")
    else: ops.set_precision("f32"); ops.set_f32_pieces(pieces)
    print(f"pieces {pieces}:   kind     N     K |  heuristic(path)   split128   split64   fp32-MFMA   [us]")
    for kind, N, K in shapes:
        out = torch.empty(M, N, device=dev)
        if kind == "nt":
            A, B = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
            f = lambda: ops.linear_nt(A, B, out=out)
        else:
            A, B = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev)
            f = lambda: ops.matmul_nn(A, B, out=out)
        row = []
        for mode in (4, 5, 7, 1):
            lib.gaot_debug_set_gemm_glds(mode)
            t = timeit(f); path = lib.gaot_debug_last_gemm_path()
            row.append(f"{t:7.1f}({path})")
        lib.gaot_debug_set_gemm_glds(4)
        print(f"             {kind} {N:6d} {K:6d} | " + "  ".join(row), flush=True)
