#!/usr/bin/env python3
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch, _exact_pairwise
dev = torch.device("cuda:0")
def t(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); return out, 1e3 * (time.perf_counter() - t0)
for (d, n, m, r) in [(2, 16384, 4096, 0.033), (2, 4096, 16384, 0.033), (3, 65536, 32768, 0.067), (3, 32768, 65536, 0.067)]:
    g = torch.Generator().manual_seed(0)
    data = (torch.rand(n, d, generator=g) * 2 - 1).to(dev); q = (torch.rand(m, d, generator=g) * 2 - 1).to(dev)
    ns = NeighborSearch()
    out, ms_hip = t(lambda: ns(data, q, r))
    _, ms_ref = t(lambda: _exact_pairwise(data, q, torch.tensor(r, device=dev), False))
    print(f"d={d} n={n} m={m} E={out['neighbors_index'].numel()} hip {ms_hip:.2f} ms  torch-exact-pairwise {ms_ref:.2f} ms", flush=True)
