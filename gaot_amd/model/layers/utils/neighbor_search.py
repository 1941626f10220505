"""Radius neighbour search -> reference-style CSR dict (neighbor_search.py:65-146 semantics: inclusive
`dist <= r`, unbounded degree, neighbours in ascending data index).

This is the BOUNDARY of the hot path: the result is cached by the MAGNO modules and consumed by the HIP
kernels through a GeometryPlan; it is not part of the steady-state step.  It runs as chunked exact pairwise distances on
whatever device the coordinates live on (a HIP cell-list builder is SURVEY 8f rank 1)."""
import torch
from torch import nn


class NeighborSearch(nn.Module):
    METHODS = ('auto', 'native', 'chunked', 'grid', 'torch_cluster', 'open3d')

    def __init__(self, method: str = 'auto', grid_size=None, chunk_size: int = 1000):
        super().__init__()
        if method not in self.METHODS:
            raise ValueError(f"unknown neighbor search method {method!r}")
        # every backend of the reference returns the same neighbour SETS for `dist <= r` except torch_cluster's
        # silent 32-neighbour cap (neighbor_search.py:163-165), which is deliberately not reproduced.
        self.method = 'native' if method in ('auto', 'grid', 'torch_cluster', 'open3d') else method
        self.chunk_size = chunk_size

    @torch.no_grad()
    def forward(self, data: torch.Tensor, queries: torch.Tensor, radius):
        r = radius if isinstance(radius, torch.Tensor) else torch.tensor(radius, device=queries.device, dtype=queries.dtype)
        per_query = isinstance(radius, torch.Tensor) and radius.dim() == 1
        if per_query and radius.numel() != queries.shape[0]:
            raise ValueError("If radius is a tensor, it must be one-dimensional and match the number of queries.")
        # exact per-pair differences (torch.cdist switches to a |q|^2+|d|^2-2qd expansion for large inputs, which
        # is fuzzy right at dist == r); bound the [chunk, n, d] difference tensor to ~256 MB
        step = max(1, min(queries.shape[0], (64 << 20) // max(1, data.shape[0] * data.shape[1])))
        cols, counts = [], []
        for s in range(0, queries.shape[0], step):
            d = (queries[s:s + step, None, :] - data[None, :, :]).square().sum(-1).sqrt()
            hit = d <= (r[s:s + step, None] if per_query else r)
            cols.append(hit.nonzero()[:, 1])
            counts.append(hit.sum(dim=1))
        index = torch.cat(cols).long()
        splits = torch.zeros(queries.shape[0] + 1, dtype=torch.long, device=queries.device)
        torch.cumsum(torch.cat(counts), dim=0, out=splits[1:])
        return {'neighbors_index': index, 'neighbors_row_splits': splits}
