#!/usr/bin/env python3
"""Do the two GEMMs of a Linear backward (input gradient NN, weight gradient TN + split-K reduce) overlap when they are
issued on two streams?  Sequential vs forked timing on the bench's FFN shapes, eager (events cost host time) and as captured
hipGraphs (fork / join are graph edges)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops
dev = torch.device("cuda:0")
side = torch.cuda.Stream()

def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(20e-3 * 2.0e9)); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

def captured(fn, reps=8):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    return timeit(g.replay, 20) / reps

for (M, N, K) in ((8192, 2048, 256), (8192, 256, 1024), (8192, 768, 256), (8192, 256, 256), (8192, 1024, 256)):
    # Linear K -> N on M rows: g [M, N], w [N, K], x [M, K]
    g = torch.randn(M, N, device=dev); w = torch.randn(N, K, device=dev); x = torch.randn(M, K, device=dev)
    dx = torch.empty(M, K, device=dev); dw = torch.empty(N, K, device=dev)
    def seq():
        ops.matmul_nn(g, w, out=dx); ops.matmul_tn(g, x, out=dw)
    def fork():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            ops.matmul_tn(g, x, out=dw)
        ops.matmul_nn(g, w, out=dx)
        cur.wait_stream(side)
    t_nn = timeit(lambda: ops.matmul_nn(g, w, out=dx)); t_tn = timeit(lambda: ops.matmul_tn(g, x, out=dw))
    print(f"Linear {K}->{N} on {M} rows: dX {t_nn:.1f} us, dW {t_tn:.1f} us | eager: sequential {timeit(seq):.1f}, two streams {timeit(fork):.1f} | "
          f"captured: sequential {captured(seq):.1f}, forked {captured(fork):.1f} us")
