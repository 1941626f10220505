#!/usr/bin/env python3
"""C2 decoder backward: dF by the row-parallel kernel inside gaot_gno_proj_backward vs the edge-partitioned kernel (batch inside
the lane group), each timed alone with HIP events.  usage: python tools/gno_c2_kernels.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib as L
from gaot_amd.ops import _p, _stream
from gaot_amd.plan import GeometryPlan
from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch
from tests._workloads import grid
dev = torch.device("cuda:0"); lib = L.load()
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(20e-3 * 2.0e9)); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
g = torch.Generator().manual_seed(0)
lat = grid([64, 64]).to(dev); x = (torch.rand(16384, 2, generator=g) * 2 - 1).to(dev)
ns = NeighborSearch("native")
d2 = ns(lat, x, 0.033)
plan = GeometryPlan(d2["neighbors_index"], d2["neighbors_row_splits"], 4096)
B, C, OC, n_src = 8, 64, 1, 4096
k = torch.randn(plan.E, C, device=dev); f = torch.randn(B, n_src, C, device=dev); weff = torch.randn(OC, C, device=dev)
dy = torch.randn(B, plan.Q, OC, device=dev); esc = torch.rand(plan.E, device=dev)
dk = torch.empty_like(k); df = torch.empty_like(f); df2 = torch.empty_like(f)
part = torch.empty(int(lib.gaot_gno_lift_edge_grad_parts(plan.E, C)), OC * C, device=dev)
ws = torch.empty(int(lib.gaot_gno_ep_workspace(plan.E, C, B)), device=dev)
def bwd(with_df):
    L.check(lib.gaot_gno_proj_backward(_p(dy), _p(k), _p(f), _p(weff), B, plan.Q, n_src, C, OC, _p(plan.index), _p(plan.edge_query), plan.E,
                                       _p(plan.t_splits), _p(plan.t_edge), _p(esc), _p(dk), _p(part), _p(df) if with_df else None, _stream()), "bwd")
def t_ep():
    L.check(lib.gaot_gno_proj_gather_t_ep(_p(k), _p(dy), _p(weff), B, plan.Q, n_src, C, OC, _p(plan.index), _p(plan.edge_query), plan.E,
                                          _p(plan.t_splits), _p(plan.t_edge), _p(esc), _p(df2), _p(ws), None, _stream()), "t_ep")
bwd(True); t_ep(); torch.cuda.synchronize()
print("max |dF row-parallel - dF edge-partitioned| / max |dF| =", float((df - df2).abs().max() / df.abs().max()))
t_all, t_edge = timeit(lambda: bwd(True)), timeit(lambda: bwd(False))
print(f"E={plan.E}: edge grad + dF (row-parallel) {t_all:.1f} us | edge grad alone {t_edge:.1f} us -> dF row-parallel {t_all - t_edge:.1f} us | dF edge-partitioned (+ fix-up) {timeit(t_ep):.1f} us")
for chunk in (4, 8, 16, 32):
    lib.gaot_debug_set_ep_chunk(chunk); ws = torch.empty(int(lib.gaot_gno_ep_workspace(plan.E, C, B)), device=dev); print(f"   chunk {chunk}: dF edge-partitioned {timeit(t_ep):.1f} us")
lib.gaot_debug_set_ep_chunk(0)
