#!/usr/bin/env python3
"""Stand-alone cost of the per-pass weight magnitude words + fp16 planes of the bench model (ops.refresh_weight_amax: one
gaot_absmax_grouped launch + one gaot_split_f16_planes_grouped launch), warm, back to back -- against their cost inside the step
(profiles/*_step_sequence.txt)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaot_amd import ops

dev = torch.device("cuda:0")
model = bench.build_model().to(dev).train()
lat, x, p, t = bench.synthetic(1234, dev)
with torch.enable_grad():
    model(latent_tokens_coord=lat, xcoord=x, pndata=p).sum().backward()      # fused groups adopt their storage
params, groups = model._amax_lists


def run(n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        ops.begin_pass()
        ops.refresh_weight_amax(params, groups)
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n


run(5)
print(f"refresh_weight_amax (absmax + planes, {len(params)} parameters, {sum(q.numel() for q in params)} values): {run(50):.1f} us per pass (host-bound if the launches are short)")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10):
        ops.begin_pass()
        ops.refresh_weight_amax(params, groups)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
g.replay(); torch.cuda.synchronize()
s.record()
for _ in range(20):
    g.replay()
e.record(); torch.cuda.synchronize()
print(f"captured, 10 passes per replay: {1e3 * s.elapsed_time(e) / 200:.1f} us per pass (incl. the arena's zero fill)")
