"""Radius neighbour search -> reference-style CSR dict (neighbor_search.py:65-146 semantics: inclusive
`dist <= r`, unbounded degree, neighbours in ascending data index like the `native` backend).

method='torch_cluster' (or max_num_neighbors=k) reproduces what the reference gets from torch_cluster.radius with its default
max_num_neighbors = 32 (neighbor_search.py:148-175): strict `d^2 < r^2` and, per query, only the k neighbours with the
smallest data indices.  The reference's 'auto' picks that backend when torch_cluster is importable and the uncapped 'grid'
backend otherwise; here 'auto' is always the uncapped exact search -- ask for 'torch_cluster' to get the capped graph.

This is the BOUNDARY of the hot path: the result is cached by the MAGNO modules and consumed by the HIP kernels
through a GeometryPlan; it is not part of the steady-state step.  On the GPU it runs the HIP cell-list builder
(gaot_cells_build / gaot_radius_count / gaot_radius_fill: O(Q * points in the 3^d surrounding cells)); host tensors
(CPU tests, dataset preparation) use exact chunked pairwise distances with the same arithmetic."""
import ctypes as C

import torch
from torch import nn


def _exact_pairwise(data: torch.Tensor, queries: torch.Tensor, r, per_query: bool, cap: int = 0, strict: bool = False):
    # exact per-pair differences (torch.cdist switches to a |q|^2+|d|^2-2qd expansion for large inputs, which
    # is fuzzy right at dist == r); bound the [chunk, n, d] difference tensor to ~256 MB
    step = max(1, min(queries.shape[0], (64 << 20) // max(1, data.shape[0] * data.shape[1])))
    cols, counts = [], []
    for s in range(0, queries.shape[0], step):
        if strict:      # torch_cluster: squared distance strictly below r^2
            hit = (data[None, :, :] - queries[s:s + step, None, :]).square().sum(-1) < (r * r)
        else:
            d = (queries[s:s + step, None, :] - data[None, :, :]).square().sum(-1).sqrt()
            hit = d <= (r[s:s + step, None] if per_query else r)
        if cap > 0:     # keep the `cap` smallest data indices of every row
            hit = hit & (hit.cumsum(dim=1) <= cap)
        cols.append(hit.nonzero()[:, 1])
        counts.append(hit.sum(dim=1))
    index = torch.cat(cols).long()
    splits = torch.zeros(queries.shape[0] + 1, dtype=torch.long, device=queries.device)
    torch.cumsum(torch.cat(counts), dim=0, out=splits[1:])
    return {'neighbors_index': index, 'neighbors_row_splits': splits}


def _hip_cell_list(data: torch.Tensor, queries: torch.Tensor, radius: float, cap: int = 0, strict: bool = False):
    from .... import _lib as L
    from ....ops import _p, _stream
    lib = L.load()
    data = data.contiguous().float()
    queries = queries.contiguous().float()
    n, dim = data.shape
    m = queries.shape[0]
    dev = data.device
    lo = data.min(dim=0).values.cpu()
    hi = data.max(dim=0).values.cpu()
    extent = float((hi - lo).max())
    # cell slightly larger than r: the +-1 cell reach then holds with margin against the rounding of (v - o) / cell
    cell = max(float(radius) * 1.001, extent / 2048.0, 1e-30)
    while True:
        dims = [int((float(hi[k]) - float(lo[k])) / cell) + 1 for k in range(dim)]
        ncell = 1
        for v in dims:
            ncell *= v
        if ncell <= (1 << 23):
            break
        cell *= 1.5
    origin = (C.c_float * dim)(*[float(lo[k]) for k in range(dim)])
    cdims = (C.c_int32 * dim)(*dims)
    cell_start = torch.empty(ncell + 1, dtype=torch.int32, device=dev)
    cell_points = torch.empty(n, dtype=torch.int32, device=dev)
    scratch = torch.empty(2 * n + ncell + 1, dtype=torch.int32, device=dev)
    L.check(lib.gaot_cells_build(_p(data), n, dim, origin, cell, cdims, _p(cell_start), _p(cell_points), _p(scratch), _stream()),
            "gaot_cells_build")
    deg = torch.empty(m, dtype=torch.int32, device=dev)
    splits = torch.empty(m + 1, dtype=torch.long, device=dev)
    L.check(lib.gaot_radius_count(_p(queries), m, _p(data), n, dim, float(radius), origin, cell, cdims, _p(cell_start),
                                  _p(cell_points), _p(deg), _p(splits), int(cap), int(strict), _stream()), "gaot_radius_count")
    E = int(splits[-1].item())
    index = torch.empty(E, dtype=torch.long, device=dev)
    if E > 0:
        sort_scratch = torch.empty(E, dtype=torch.long, device=dev)
        L.check(lib.gaot_radius_fill(_p(queries), m, _p(data), n, dim, float(radius), origin, cell, cdims, _p(cell_start),
                                     _p(cell_points), _p(splits), _p(index), _p(sort_scratch), int(cap), int(strict), _stream()),
                "gaot_radius_fill")
    return {'neighbors_index': index, 'neighbors_row_splits': splits}


class NeighborSearch(nn.Module):
    METHODS = ('auto', 'native', 'chunked', 'grid', 'torch_cluster', 'open3d')

    def __init__(self, method: str = 'auto', grid_size=None, chunk_size: int = 1000, max_num_neighbors=None):
        super().__init__()
        if method not in self.METHODS:
            raise ValueError(f"unknown neighbor search method {method!r}")
        # native / chunked / grid / open3d return the same neighbour SETS for `dist <= r`; torch_cluster caps every query at
        # max_num_neighbors (default 32) and tests `d^2 < r^2`
        self.requested_method = method
        self.method = 'torch_cluster' if method == 'torch_cluster' else ('native' if method in ('auto', 'grid', 'open3d') else method)
        self.max_num_neighbors = int(max_num_neighbors) if max_num_neighbors is not None else (32 if method == 'torch_cluster' else 0)
        self.strict = method == 'torch_cluster'
        self.chunk_size = chunk_size

    @torch.no_grad()
    def forward(self, data: torch.Tensor, queries: torch.Tensor, radius):
        per_query = isinstance(radius, torch.Tensor) and radius.dim() == 1
        if per_query and radius.numel() != queries.shape[0]:
            raise ValueError("If radius is a tensor, it must be one-dimensional and match the number of queries.")
        cap, strict = self.max_num_neighbors, self.strict
        if data.is_cuda and not per_query and data.shape[1] in (2, 3) and data.shape[0] > 0 and queries.shape[0] > 0:
            return _hip_cell_list(data, queries, float(radius), cap, strict)
        r = radius if isinstance(radius, torch.Tensor) else torch.tensor(radius, device=queries.device, dtype=queries.dtype)
        return _exact_pairwise(data, queries, r, per_query, cap, strict)
