"""GeometryPlan: everything the GNO kernels need that depends only on the mesh geometry.

The reference recomputes these per forward with repeat_interleave / torch_scatter (agno.py:188-224,
gemb.py:103-171); here they are device arrays built once per neighbour list and cached on it:
  index / splits (int32 CSR), edge_query (edge -> query id), transposed CSR (t_splits, t_edge) for the
  scatter-free backward, 1/deg per edge ('mean' reduction), and -- per coordinate pair -- the kernel-MLP
  input rows, the cosine attention weights and the standardised geometry statistics.
"""
import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib as L
from .ops import _p, _stream

_PLAN_KEY = "_gaot_amd_plan"


class GeometryPlan:
    def __init__(self, index_i64: torch.Tensor, splits_i64: torch.Tensor, n_src: int, validate: bool = True):
        if not index_i64.is_cuda or not splits_i64.is_cuda:
            raise RuntimeError("GeometryPlan needs the CSR on the GPU (gaot_amd has no CPU path)")
        lib = L.load()
        dev = splits_i64.device
        index_i64 = index_i64.contiguous().long()
        splits_i64 = splits_i64.contiguous().long()
        self.Q = int(splits_i64.numel() - 1)
        self.E = int(index_i64.numel())
        self.n_src = int(n_src)
        E1 = max(self.E, 1)
        self.index = torch.empty(E1, dtype=torch.int32, device=dev)
        self.splits = torch.empty(self.Q + 1, dtype=torch.int32, device=dev)
        self.edge_query = torch.empty(E1, dtype=torch.int32, device=dev)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        L.check(lib.gaot_csr_prepare(_p(index_i64), _p(splits_i64), self.Q, self.E, self.n_src, _p(self.index),
                                     _p(self.splits), _p(self.edge_query), _p(flag), _stream()), "gaot_csr_prepare")
        self.t_splits = torch.empty(self.n_src + 1, dtype=torch.int32, device=dev)
        self.t_edge = torch.empty(E1, dtype=torch.int32, device=dev)
        scratch = torch.empty(self.n_src + 1, dtype=torch.int32, device=dev)
        L.check(lib.gaot_csr_transpose(_p(self.index), self.E, self.n_src, _p(self.t_splits), _p(self.t_edge), _p(scratch),
                                       _stream()), "gaot_csr_transpose")
        bad = int(flag.item()) if validate else 0     # one sync per geometry: the CSR contract is checked on device
        if bad:
            raise ValueError(f"invalid CSR neighbour list (flag {bad}: 1 = row_splits not monotone 0..E, 2 = index outside [0, n_src))")
        deg = (splits_i64[1:] - splits_i64[:-1])
        self.deg = deg
        self._inv_deg_edge: Optional[torch.Tensor] = None
        self._edge_query_long: Optional[torch.Tensor] = None
        self._index_long = index_i64
        self._coord_cache: Dict[str, dict] = {}
        self.epoch = 0                # bumped whenever a coordinate-derived array may have been refreshed in place

    # ---- lazily derived
    @property
    def edge_query_long(self) -> torch.Tensor:
        if self._edge_query_long is None:
            self._edge_query_long = self.edge_query[:self.E].long()
        return self._edge_query_long

    @property
    def index_long(self) -> torch.Tensor:
        return self._index_long

    @property
    def inv_deg_edge(self) -> torch.Tensor:
        """1/deg(query(e)) per edge -- the 'mean' reduction of agno.py:264 as a per-edge scale."""
        if self._inv_deg_edge is None:
            inv = 1.0 / self.deg.clamp(min=1).to(torch.float32)
            self._inv_deg_edge = inv[self.edge_query_long].contiguous() if self.E > 0 else inv.new_zeros(1)
        return self._inv_deg_edge

    def _cached(self, name: str, tensors, alloc, compute):
        """geometry-only array `name` derived from coordinate tensors.
          * same tensor objects (and versions) as last time            -> the cached array, no launch;
          * NEW objects of the same shape (a trainer that re-uploads the coordinates every step,
            static_trainer.py:167-170)                                  -> device-side content guard: the bytes are compared
            with the kept copy and the array is recomputed IN PLACE only if they differ -- no host synchronisation;
          * otherwise                                                   -> fresh allocation and computation."""
        key = tuple((id(t), t._version) for t in tensors)
        hit = self._coord_cache.get(name)
        if hit is not None and hit["key"] == key:
            return hit["val"]
        lib = L.load()
        if hit is not None and all(t.shape == k.shape and t.dtype == k.dtype and t.device == k.device for t, k in zip(tensors, hit["kept"])):
            flag = hit["flag"]
            cur = [t.contiguous() for t in tensors]
            L.check(lib.gaot_guard_begin(_p(flag), _stream()), "gaot_guard_begin")
            for t, k in zip(cur, hit["kept"]):
                L.check(lib.gaot_guard_compare(_p(t), _p(k), t.numel() * t.element_size(), _p(flag), _stream()), "gaot_guard_compare")
            for t, k in zip(cur, hit["kept"]):
                L.check(lib.gaot_guard_update(_p(t), _p(k), t.numel() * t.element_size(), _p(flag), _stream()), "gaot_guard_update")
            compute(hit["full"], cur, flag)
            hit["key"], hit["hold"] = key, tuple(tensors)
            self.epoch += 1           # dependants cached on the host (inference-time kernel values, row bias) must re-derive
            return hit["val"]
        cur = [t.contiguous() for t in tensors]
        full, val = alloc(cur)
        compute(full, cur, None)
        self._coord_cache[name] = {"key": key, "hold": tuple(tensors), "val": val, "full": full,
                                   "kept": [t.clone() for t in cur],
                                   "flag": torch.zeros(1, dtype=torch.int32, device=cur[0].device)}
        return val

    def edge_features(self, src: torch.Tensor, qry: torch.Tensor) -> torch.Tensor:
        """[y_j, x_i] rows of the kernel MLP (agno.py:229)."""
        def alloc(ts):
            feat = torch.empty(max(self.E, 1), 2 * ts[0].shape[1], device=ts[0].device, dtype=torch.float32)
            return feat, feat[:self.E]

        def compute(feat, ts, guard):
            L.check(L.load().gaot_edge_features(_p(ts[0]), _p(ts[1]), ts[0].shape[1], _p(self.index), _p(self.edge_query), self.E,
                                                _p(feat), _p(guard), _stream()), "gaot_edge_features")
        return self._cached("feat", (src, qry), alloc, compute)

    def cosine_attention(self, src: torch.Tensor, qry: torch.Tensor) -> torch.Tensor:
        def alloc(ts):
            attn = torch.zeros(max(self.E, 1), device=ts[0].device, dtype=torch.float32)
            return attn, attn

        def compute(attn, ts, guard):
            L.check(L.load().gaot_edge_attention_cosine(_p(ts[0]), _p(ts[1]), ts[0].shape[1], _p(self.index), _p(self.splits), self.Q,
                                                        _p(attn), _p(guard), _stream()), "gaot_edge_attention_cosine")
        return self._cached("cos", (src, qry), alloc, compute)

    def geo_stats(self, geom: torch.Tensor, qry: torch.Tensor) -> torch.Tensor:
        def alloc(ts):
            F = 3 + 2 * ts[0].shape[1]
            stats = torch.empty(self.Q, F, device=ts[0].device, dtype=torch.float32)
            return (stats, torch.empty(4 * F, device=ts[0].device, dtype=torch.float64)), stats

        def compute(full, ts, guard):
            stats, scratch = full
            L.check(L.load().gaot_geo_stats(_p(ts[0]), _p(ts[1]), ts[0].shape[1], _p(self.index), _p(self.splits), self.Q, _p(stats),
                                            _p(scratch), _p(guard), _stream()), "gaot_geo_stats")
        return self._cached("stats", (geom, qry), alloc, compute)


def plan_for(neighbors: dict, n_src: int) -> GeometryPlan:
    """Plan attached to (and cached on) a reference-style neighbour dict."""
    plan = neighbors.get(_PLAN_KEY)
    idx = neighbors["neighbors_index"]
    if plan is None or plan._src_id != (id(idx), idx._version) or plan.n_src != n_src:
        plan = GeometryPlan(idx, neighbors["neighbors_row_splits"], n_src)
        plan._src_id = (id(idx), idx._version)
        neighbors[_PLAN_KEY] = plan
    return plan


class MergedGeometry:
    """Block-diagonal union of per-sample geometries (vx mode, reference magno.py:356-413 / 694-751 loops over
    samples in Python): sources and queries of all samples are concatenated, CSR indices are offset per sample, so
    the whole minibatch goes through ONE launch of each GNO kernel with a batch dimension of 1.

    Per-sample quantities that the reference normalises per geometry (the geometry statistics' global
    standardisation, gemb.py:164-169) are computed per sample and concatenated."""

    def __init__(self, nbr_dicts, src_coords, dst_coords):
        idx, sp, self.n_src_each, self.n_dst_each = [], [], [], []
        e_off = s_off = 0
        dev = nbr_dicts[0]["neighbors_row_splits"].device
        for nb, sc, dc in zip(nbr_dicts, src_coords, dst_coords):
            i, s_ = nb["neighbors_index"], nb["neighbors_row_splits"]
            idx.append(i + s_off)
            sp.append(s_[:-1] + e_off)
            e_off += int(i.numel())
            s_off += int(sc.shape[0])
            self.n_src_each.append(int(sc.shape[0]))
            self.n_dst_each.append(int(dc.shape[0]))
        sp.append(torch.tensor([e_off], dtype=torch.long, device=dev))
        self.neighbors = {"neighbors_index": torch.cat(idx), "neighbors_row_splits": torch.cat(sp)}
        self.src = torch.cat(list(src_coords), dim=0).contiguous()
        self.dst = torch.cat(list(dst_coords), dim=0).contiguous()
        for nb, sc in zip(nbr_dicts, src_coords):          # validates each part once (cached on the per-sample dict)
            plan_for(nb, sc.shape[0])
        plan = GeometryPlan(self.neighbors["neighbors_index"], self.neighbors["neighbors_row_splits"], self.src.shape[0],
                            validate=False)                # parts are valid => the offset union is valid: no host sync
        plan._src_id = (id(self.neighbors["neighbors_index"]), self.neighbors["neighbors_index"]._version)
        self.neighbors[_PLAN_KEY] = plan
        self.plan = plan
        self._parts = (list(nbr_dicts), list(src_coords), list(dst_coords))
        self._stats = None

    def geo_stats(self) -> torch.Tensor:
        if self._stats is None:
            nbs, scs, dcs = self._parts
            self._stats = torch.cat([plan_for(nb, sc.shape[0]).geo_stats(sc, dc) for nb, sc, dc in zip(nbs, scs, dcs)], dim=0)
        return self._stats


_MERGE_CACHE = {}


def merged_geometry(nbr_dicts, src_coords, dst_coords, parents=()) -> MergedGeometry:
    """Cached on the identity of the per-sample neighbour dicts and of the PARENT coordinate tensors (per-sample
    slices are new objects on every call).  A re-shuffled batch is a new combination and is merged afresh."""
    key = tuple(id(n) for n in nbr_dicts) + tuple((id(c), c._version) for c in parents)
    hit = _MERGE_CACHE.get(key)
    if hit is None or any(a is not b for a, b in zip(hit[1], parents)):
        if len(_MERGE_CACHE) > 64:
            _MERGE_CACHE.clear()
        hit = (MergedGeometry(nbr_dicts, src_coords, dst_coords), tuple(parents), list(nbr_dicts))   # hold refs: ids stay unique
        _MERGE_CACHE[key] = hit
    return hit[0]
