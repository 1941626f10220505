#!/usr/bin/env python3
"""Edge-partitioned transform kernels: time vs chunk size on the C3 geometry (merged union, B = 1) and the C2 geometry (B = 8).
usage: python tools/gno_ep_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib as L
from gaot_amd.plan import GeometryPlan, MergedGeometry
from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch
from tests._workloads import grid, naca_points
dev = torch.device("cuda:0"); lib = L.load()
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
g = torch.Generator().manual_seed(0)
lat = grid([64, 64]).to(dev)
ns = NeighborSearch("native")
for name, B, npts, ci in (("C3 union of 16 skewed samples", 1, 8192, 3), ("C2 uniform", 8, 16384, 1)):
    if B == 1:
        xs = [naca_points(npts, g, 0.15).to(dev) for _ in range(16)]
        enc = MergedGeometry([ns(x, lat, 0.033) for x in xs], xs, [lat] * 16).plan
        dec = MergedGeometry([ns(lat, x, 0.033) for x in xs], [lat] * 16, xs).plan
        n_src_e, n_src_d = 16 * npts, 16 * 4096
    else:
        x = (torch.rand(npts, 2, generator=g) * 2 - 1).to(dev)
        d1, d2 = ns(x, lat, 0.033), ns(lat, x, 0.033)
        enc, dec = GeometryPlan(d1["neighbors_index"], d1["neighbors_row_splits"], npts), GeometryPlan(d2["neighbors_index"], d2["neighbors_row_splits"], 4096)
        n_src_e, n_src_d = npts, 4096
    C = 64
    k_e, k_d = torch.randn(enc.E, C, device=dev), torch.randn(dec.E, C, device=dev)
    pn = torch.randn(B, n_src_e, ci, device=dev); wl = torch.randn(C, ci, device=dev); bl = torch.randn(C, device=dev)
    a_e, a_d = torch.rand(enc.E, device=dev), torch.rand(dec.E, device=dev)
    f = torch.randn(B, n_src_d, C, device=dev); weff = torch.randn(1, C, device=dev)
    dy = torch.randn(B, dec.Q, 1, device=dev)
    bytes_e = 4.0 * (enc.E * C + B * n_src_e * ci + 3 * enc.E + B * enc.Q * C)
    bytes_t = 4.0 * (dec.E * C + B * dec.Q + 4 * dec.E + B * n_src_d * C)
    print(f"== {name}: encoder E={enc.E} Q={enc.Q}, decoder E={dec.E} Q={dec.Q}")
    for mode, chunk in ((0, 0), (1, 8), (1, 16), (1, 32), (1, 64)):
        ops.set_gno_ep(mode); lib.gaot_debug_set_ep_chunk(chunk)
        with torch.no_grad():
            us_e = timeit(lambda: ops._GNOLiftTransform.apply(k_e, pn, wl, bl, enc, a_e))
        kd = k_d.clone().requires_grad_(True); ff = f.clone().requires_grad_(True)
        y = ops.gno_proj_transform(kd, ff, weff, None, None, dec, a_d)
        us_bwd = timeit(lambda: torch.autograd.grad(y, [kd, ff], dy, retain_graph=True))
        with torch.no_grad():
            us_f = timeit(lambda: ops.gno_proj_transform(k_d, f, weff, None, None, dec, a_d))
        print(f"  mode {mode} chunk {chunk:3d}: encoder fwd {us_e:7.1f} us ({bytes_e / us_e / 1e3:6.0f} GB/s) | decoder fwd {us_f:7.1f} us | decoder bwd (edge grad + dF [+ colsum]) {us_bwd:7.1f} us")
    lib.gaot_debug_set_ep_chunk(0); ops.set_gno_ep(2)
