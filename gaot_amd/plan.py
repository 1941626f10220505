"""GeometryPlan: everything the GNO kernels need that depends only on the mesh geometry.

The reference recomputes these per forward with repeat_interleave / torch_scatter (agno.py:188-224,
gemb.py:103-171); here they are device arrays built once per neighbour list and cached on it:
  index / splits (int32 CSR), edge_query (edge -> query id), transposed CSR (t_splits, t_edge) for the
  scatter-free backward, 1/deg per edge ('mean' reduction), and -- per coordinate pair -- the kernel-MLP
  input rows, the cosine attention weights and the standardised geometry statistics.
"""
import ctypes as C
import itertools
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import os
import numpy as np
import torch

from . import _lib as L
from .ops import _p, _stream

_PLAN_KEY = "_gaot_amd_plan"


_PENDING: list = []          # (device flag, event) of plans built with validate="lazy"
GEO_SCRATCH = 34             # doubles per (group, statistic) of gaot_geo_stats' scratch (include/gaot_hip.h)
FORCE_GUARD = [None]         # set by autograph.py while it captures: a device flag that guards re-computation inside the graph


def _raise_if_bad(bad: int):
    if bad:
        raise ValueError(f"invalid CSR neighbour list (flag {bad}: 1 = row_splits not monotone 0..E, 2 = index outside [0, n_src))")


def _check_pending(block: bool = False):
    keep = []
    for flag, ev in _PENDING:
        if block or ev.query():
            _raise_if_bad(int(flag.item()))
        else:
            keep.append((flag, ev))
    _PENDING[:] = keep


class GeometryPlan:
    e_dev: Optional[torch.Tensor] = None      # padded unions only (StaticUnion): device scalar = the number of edges actually in the list (<= E)

    def __init__(self, index_i64: torch.Tensor, splits_i64: torch.Tensor, n_src: int, validate=True):
        if index_i64 is None:          # assembled by GeometryPlan.compose()
            return
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
        # the CSR contract is checked on device.  validate=True: one host sync per geometry, BEFORE anything is derived from the
        # list; "lazy": the flag is read when a later plan is built (per-step geometries of a vx batch: no sync on the step that
        # uploads them, the error surfaces one step on; csr_prepare clamps what it stores so the kernels in between stay in bounds)
        _check_pending()
        if validate == "lazy":
            ev = torch.cuda.Event()
            ev.record()
            _PENDING.append((flag, ev))
        elif validate:
            _raise_if_bad(int(flag.item()))
        self.t_splits = torch.empty(self.n_src + 1, dtype=torch.int32, device=dev)
        self.t_edge = torch.empty(E1, dtype=torch.int32, device=dev)
        scratch = torch.empty(self.n_src + 1 + E1, dtype=torch.int32, device=dev)
        L.check(lib.gaot_csr_transpose(_p(self.index), self.E, self.n_src, _p(self.t_splits), _p(self.t_edge), _p(scratch),
                                       _stream()), "gaot_csr_transpose")
        deg = (splits_i64[1:] - splits_i64[:-1])
        self.deg = deg
        # degree skew decides between the row-parallel and the edge-partitioned transform kernels (ops.py).  A plan built with a
        # host sync anyway (validate=True) reads the two maxima now; a lazily validated one does not wait for them and is
        # treated as skewed (the edge-partitioned kernels are right for any distribution)
        self.max_deg = self.max_t_deg = (0 if self.E == 0 else None)
        if validate is True and self.E > 0:
            td = self.t_splits[1:] - self.t_splits[:-1]
            m = torch.stack([deg.max().to(torch.int64), td.max().to(torch.int64)]).cpu()
            self.max_deg, self.max_t_deg = int(m[0]), int(m[1])
        self._inv_deg_edge: Optional[torch.Tensor] = None
        self._edge_query_long: Optional[torch.Tensor] = None
        self._index_long = index_i64
        self._coord_cache: Dict[str, dict] = {}
        self._groups: Dict[tuple, dict] = {}          # per coordinate pair (by shapes): kept bytes + the arrays derived from them
        self.epoch = 0                # bumped whenever a coordinate-derived array may have been refreshed in place

    SKEW_DEGREE = 48      # rows longer than this (or unknown) go to the edge-partitioned kernels

    def part_pointers(self):
        """(index, edge_query, t_edge, splits, t_splits) device addresses: this plan as one gaot_union_part of a StaticUnion's table"""
        pp = getattr(self, "_part_ptrs", None)
        if pp is None:
            pp = self._part_ptrs = (self.index.data_ptr(), self.edge_query.data_ptr(), self.t_edge.data_ptr(), self.splits.data_ptr(), self.t_splits.data_ptr())
        return pp

    @property
    def rows_skewed(self) -> bool:
        return self.max_deg is None or self.max_deg > self.SKEW_DEGREE

    @property
    def t_rows_skewed(self) -> bool:
        return self.max_t_deg is None or self.max_t_deg > self.SKEW_DEGREE

    # ---- lazily derived
    @property
    def row_order(self) -> Optional[torch.Tensor]:
        """the rows sorted by their first source row (int32 [Q]; empty rows last), for kernels that gather source-row features per
        row: rows that share neighbours then share workgroups and L2s (csrc/gno_ep.hip proj_fwd_bin_kernel).  Only for plans that
        live across steps (built with a host sync: the dict-cached geometries of fx mode) and are worth it; None otherwise."""
        ro = getattr(self, "_row_order", None)
        if ro is None and getattr(self, "_src_id", None) is not None and getattr(self, "max_deg", None) is not None and self.E > 0 and self.Q >= 2048 and not torch.cuda.is_current_stream_capturing():
            first = self.index_long[self.splits[:-1].long().clamp(max=self.E - 1)]
            key = torch.where(self.deg > 0, first, torch.full_like(first, self.n_src))
            ro = self._row_order = torch.argsort(key, stable=True).to(torch.int32)
        return ro

    @property
    def edge_query_long(self) -> torch.Tensor:
        if self._edge_query_long is None:
            self._edge_query_long = self.edge_query[:self.E].long()
        return self._edge_query_long

    @property
    def index_long(self) -> torch.Tensor:
        if self._index_long is None:
            self._index_long = self.index[:self.E].long()
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
          * NEW objects of the same shapes (a trainer that re-uploads the coordinates every step,
            static_trainer.py:167-170)                                  -> device-side content guard: the bytes are compared
            with the kept copy ONCE for all arrays derived from that coordinate pair, and those arrays are recomputed IN PLACE
            only if they differ -- no host synchronisation;
          * otherwise                                                   -> fresh allocation and computation."""
        key = tuple((id(t), t._version) for t in tensors)
        hit = self._coord_cache.get(name)
        if hit is not None and hit["key"] == key:
            if FORCE_GUARD[0] is not None:
                # autograph.py is capturing: the refresh of this array becomes part of the captured forward, guarded by a flag
                # the caller raises when it finds new coordinate bytes (the tensors here are its static coordinate buffers)
                cur = [t.contiguous() for t in tensors]
                compute(hit["full"], cur, FORCE_GUARD[0])
                grp = hit["group"]
                if next(iter(grp["arrays"].values())) is hit:
                    # ... and so does the update of the pair's kept bytes (once per pair): the eager content guard compares
                    # against what the arrays were last derived from, whoever refreshed them
                    for t, k in zip(cur, grp["kept"]):
                        L.check(L.load().gaot_guard_update(_p(t), _p(k), t.numel() * t.element_size(), _p(FORCE_GUARD[0]), _stream()),
                                "gaot_guard_update")
            return hit["val"]
        gkey = tuple((tuple(t.shape), t.dtype, t.device) for t in tensors)
        grp = self._groups.get(gkey)
        cur = [t.contiguous() for t in tensors]
        if grp is not None and grp["key"] != key:
            self._refresh_group(grp, key, tensors, cur)
        if hit is not None and hit.get("group") is grp and grp is not None:
            return hit["val"]
        full, val = alloc(cur)
        compute(full, cur, None)
        if grp is None:
            grp = {"key": key, "hold": tuple(tensors), "kept": [t.clone() for t in cur], "arrays": {},
                   "flag": torch.zeros(1, dtype=torch.int32, device=cur[0].device)}
            self._groups[gkey] = grp
        entry = {"key": key, "val": val, "full": full, "compute": compute, "group": grp}
        grp["arrays"][name] = entry
        self._coord_cache[name] = entry
        return val

    def _refresh_group(self, grp, key, tensors, cur):
        """new tensor objects for a coordinate pair the plan already holds arrays for: compare, update the kept copy, recompute
        every array of the pair under the guard flag (all of them: they must stay consistent with the kept bytes)"""
        lib = L.load()
        flag = grp["flag"]
        L.check(lib.gaot_guard_begin(_p(flag), _stream()), "gaot_guard_begin")
        for t, k in zip(cur, grp["kept"]):
            L.check(lib.gaot_guard_compare(_p(t), _p(k), t.numel() * t.element_size(), _p(flag), _stream()), "gaot_guard_compare")
        for t, k in zip(cur, grp["kept"]):
            L.check(lib.gaot_guard_update(_p(t), _p(k), t.numel() * t.element_size(), _p(flag), _stream()), "gaot_guard_update")
        for entry in grp["arrays"].values():
            entry["compute"](entry["full"], cur, flag)
            entry["key"] = key
        grp["key"], grp["hold"] = key, tuple(tensors)
        self.epoch += 1               # dependants cached on the host (inference-time kernel values, row bias) must re-derive

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

    def geo_stats(self, geom: torch.Tensor, qry: torch.Tensor, groups: int = 1) -> torch.Tensor:
        """`groups`: the query rows are that many equal groups, each standardised on its own (vx mode: one per sample)"""
        def alloc(ts):
            F = 3 + 2 * ts[0].shape[1]
            stats = torch.empty(self.Q, F, device=ts[0].device, dtype=torch.float32)
            return (stats, torch.empty(GEO_SCRATCH * F * groups, device=ts[0].device, dtype=torch.float64)), stats

        def compute(full, ts, guard):
            stats, scratch = full
            L.check(L.load().gaot_geo_stats(_p(ts[0]), _p(ts[1]), ts[0].shape[1], _p(self.index), _p(self.splits), self.Q, _p(stats),
                                            _p(scratch), _p(guard), groups, _stream()), "gaot_geo_stats")
        return self._cached(f"stats{groups}", (geom, qry), alloc, compute)


def plan_for(neighbors: dict, n_src: int, validate=True) -> GeometryPlan:
    """Plan attached to (and cached on) a reference-style neighbour dict.  `validate`: as GeometryPlan (used when the plan is built)."""
    plan = neighbors.get(_PLAN_KEY)
    if plan is not None and plan._src_id is None:          # the union of a MergedGeometry: owned by it, nothing to re-derive
        return plan
    idx = neighbors["neighbors_index"]
    if plan is None or plan._src_id != (id(idx), idx._version) or plan.n_src != n_src:
        plan = GeometryPlan(idx, neighbors["neighbors_row_splits"], n_src, validate=validate)
        plan._src_id = (id(idx), idx._version)
        neighbors[_PLAN_KEY] = plan
    return plan


def has_plan(neighbors: dict, n_src: int) -> bool:
    """whether plan_for(neighbors, n_src) would return a cached plan (nothing to build)"""
    plan = neighbors.get(_PLAN_KEY)
    if plan is None:
        return False
    if plan._src_id is None:
        return True
    idx = neighbors["neighbors_index"]
    return plan._src_id == (id(idx), idx._version) and plan.n_src == n_src


_RENUM_KEY = "_gaot_amd_renumbered"


def renumbered(neighbors: dict, role: str, perm: torch.Tensor, inv: torch.Tensor) -> dict:
    """The caller's graph over a RENUMBERED latent grid: latent point perm[r] of the caller becomes point r (inv[perm[r]] = r).
    model/gaot.py runs encoder and decoder on the latent grid in patch-major order, which turns the processor's patchify / unpatchify
    permutes (gaot.py:182-186, 222-229) into reshapes.  role 'queries' (encoder: the latent points are the CSR's rows -- rows taken in `perm`
    order, every row keeps its edges in the caller's order) or 'sources' (decoder: the latent points are what neighbors_index names --
    relabelled through `inv`, rows and edge order untouched).  Every per-row reduction sums the same terms in the same order as over the
    caller's list.  Cached on the caller's dict for as long as its two tensors are the same objects at the same versions."""
    idx, sp = neighbors["neighbors_index"], neighbors["neighbors_row_splits"]
    key = (role, id(idx), idx._version, id(sp), sp._version, id(perm))
    hit = neighbors.get(_RENUM_KEY)
    if hit is not None and hit[0] == key and hit[2] is perm:
        return hit[1]
    if role == "sources":
        out = {"neighbors_index": inv[idx.long()].to(idx.dtype), "neighbors_row_splits": sp}
    elif role == "queries":
        spl = sp.long()
        Q, E = int(spl.numel() - 1), int(idx.numel())
        if Q != int(perm.numel()):
            raise ValueError(f"renumbered: {Q} CSR rows for a latent grid of {int(perm.numel())} points")
        cnt = (spl[1:] - spl[:-1])[perm]
        nsp = torch.zeros_like(spl)
        nsp[1:] = torch.cumsum(cnt, 0)
        rows = torch.repeat_interleave(torch.arange(Q, device=spl.device), cnt, output_size=E)
        pos = spl[:-1][perm][rows] + (torch.arange(E, device=spl.device) - nsp[:-1][rows])
        out = {"neighbors_index": idx[pos], "neighbors_row_splits": nsp.to(sp.dtype)}
    else:
        raise ValueError(f"renumbered: role {role!r}")
    neighbors[_RENUM_KEY] = (key, out, perm, idx, sp)          # (hold the tensors: their ids stay unique)
    return out


def _i32_array(vals):
    arr = (C.c_int32 * len(vals))()
    for i, v in enumerate(vals):
        arr[i] = int(v)
    return arr


def _concat_offset(tensors, lens, offsets, out):
    ptrs = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        ptrs[i] = t.data_ptr()
    L.check(L.load().gaot_concat_offset(ptrs, _i32_array(lens), _i32_array(offsets), len(tensors), _p(out), _stream()), "gaot_concat_offset")


def compose_plans(plans) -> GeometryPlan:
    """Block-diagonal union of already built per-sample plans WITHOUT re-deriving anything: sources and queries of the samples
    are stacked, so every int32 array of the union is the concatenation of the per-sample arrays plus a per-sample offset --
    including the transposed CSR (a source belongs to exactly one sample, so its edge list is unchanged up to the edge offset,
    and stays sorted).  Five small launches instead of a count / scan / fill / sort pipeline over the union."""
    dev = plans[0].splits.device
    m = GeometryPlan(None, None, 0)
    m.Q, m.E, m.n_src = sum(p.Q for p in plans), sum(p.E for p in plans), sum(p.n_src for p in plans)
    E1 = max(m.E, 1)
    e_off, q_off, s_off = [0], [0], [0]
    for p in plans:
        e_off.append(e_off[-1] + p.E); q_off.append(q_off[-1] + p.Q); s_off.append(s_off[-1] + p.n_src)
    m.index = torch.empty(E1, dtype=torch.int32, device=dev)
    m.edge_query = torch.empty(E1, dtype=torch.int32, device=dev)
    m.t_edge = torch.empty(E1, dtype=torch.int32, device=dev)
    m.splits = torch.empty(m.Q + 1, dtype=torch.int32, device=dev)
    m.t_splits = torch.empty(m.n_src + 1, dtype=torch.int32, device=dev)
    Es = [p.E for p in plans]
    if m.E > 0:
        _concat_offset([p.index for p in plans], Es, s_off[:-1], m.index)
        _concat_offset([p.edge_query for p in plans], Es, q_off[:-1], m.edge_query)
        _concat_offset([p.t_edge for p in plans], Es, e_off[:-1], m.t_edge)
    # row splits: every part contributes its Q (n_src) leading entries; the last part also its closing entry
    last = len(plans) - 1
    _concat_offset([p.splits for p in plans], [p.Q + (1 if i == last else 0) for i, p in enumerate(plans)], e_off[:-1], m.splits)
    _concat_offset([p.t_splits for p in plans], [p.n_src + (1 if i == last else 0) for i, p in enumerate(plans)], e_off[:-1], m.t_splits)
    m.deg = torch.cat([p.deg for p in plans])
    m.max_deg = None if any(p.max_deg is None for p in plans) else max(p.max_deg for p in plans)
    m.max_t_deg = None if any(p.max_t_deg is None for p in plans) else max(p.max_t_deg for p in plans)
    m._inv_deg_edge = m._edge_query_long = None
    m._index_long = None
    m._coord_cache = {}
    m._groups = {}
    m.epoch = 0
    return m


class MergedGeometry:
    """Block-diagonal union of per-sample geometries (vx mode, reference magno.py:356-413 / 694-751 loops over
    samples in Python): sources and queries of all samples are concatenated, CSR indices are offset per sample, so
    the whole minibatch goes through ONE launch of each GNO kernel with a batch dimension of 1.

    Two ways to get the union's plan:
      * every per-sample neighbour dict already carries a plan (dicts kept on the device across steps, any batch order):
        `compose_plans` -- concatenation with offsets, no counting sort, no per-row sort, no host synchronisation;
      * otherwise (a loader that uploads fresh dicts every step): one plan over the concatenated CSR, validated lazily.
    Per-sample quantities that the reference normalises per geometry (the geometry statistics' global standardisation,
    gemb.py:164-169) are standardised per sample group inside one launch."""

    def __init__(self, nbr_dicts, src_coords, dst_coords, build_parts: bool = False):
        B = len(nbr_dicts)
        self.B = B
        self.n_src_each = [int(c.shape[0]) for c in src_coords]
        self.n_dst_each = [int(c.shape[0]) for c in dst_coords]
        self.src = torch.cat(list(src_coords), dim=0).contiguous()
        self.dst = torch.cat(list(dst_coords), dim=0).contiguous()
        if build_parts:
            for nb, n in zip(nbr_dicts, self.n_src_each):
                plan_for(nb, n)
        parts = [nb.get(_PLAN_KEY) for nb in nbr_dicts]
        ok = all(p is not None and p.n_src == n and p._src_id == (id(nb["neighbors_index"]), nb["neighbors_index"]._version)
                 for p, n, nb in zip(parts, self.n_src_each, nbr_dicts))
        if ok:
            plan = compose_plans(parts)
            self.composed = True
        else:
            dev = nbr_dicts[0]["neighbors_row_splits"].device
            Es = [int(nb["neighbors_index"].numel()) for nb in nbr_dicts]
            s_off = torch.tensor([sum(self.n_src_each[:b]) for b in range(B)], dtype=torch.long).to(dev, non_blocking=True)
            e_off = torch.tensor([sum(Es[:b]) for b in range(B)], dtype=torch.long).to(dev, non_blocking=True)
            rep = torch.tensor(Es, dtype=torch.long).to(dev, non_blocking=True)
            idx = torch.cat([nb["neighbors_index"] for nb in nbr_dicts]) + torch.repeat_interleave(s_off, rep, output_size=sum(Es))
            nd = torch.tensor(self.n_dst_each, dtype=torch.long).to(dev, non_blocking=True)
            sp = torch.cat([nb["neighbors_row_splits"][:-1] for nb in nbr_dicts]) + torch.repeat_interleave(e_off, nd, output_size=sum(self.n_dst_each))
            sp = torch.cat([sp, torch.full((1,), sum(Es), dtype=torch.long, device=dev)])
            plan = GeometryPlan(idx, sp, self.src.shape[0], validate="lazy")
            self.composed = False
        self.neighbors = {"neighbors_index": None, "neighbors_row_splits": None, _PLAN_KEY: plan}
        plan._src_id = None
        self.plan = plan
        self._equal = len(set(self.n_dst_each)) == 1

    def release(self) -> None:
        """Drop the derived arrays of a union that left the cache.  They sit in reference cycles (an array's entry keeps the closure that
        recomputes it, which keeps the plan; entry and coordinate group point at each other), so without this the device memory of an
        evicted union waits for Python's cyclic collector -- generations later.  Anything still using the plan recomputes on demand."""
        plan = self.plan
        for grp in plan._groups.values():
            grp["arrays"].clear()
        plan._groups.clear()
        plan._coord_cache.clear()

    def geo_stats(self) -> torch.Tensor:
        if not self._equal:
            raise ValueError("vx mode needs the same number of query points in every sample of a batch")
        return self.plan.geo_stats(self.src, self.dst, groups=self.B)


_MERGE_CACHE = {}
_MERGE_CACHE_MAX = max(2, int(os.environ.get("GAOT_MERGE_CACHE", "8")))          # unions kept (encoder and decoder count separately)


def merged_geometry(nbr_dicts, src_coords, dst_coords, parents=()) -> MergedGeometry:
    """Cached on the identity of the per-sample neighbour dicts and of the PARENT coordinate tensors (per-sample
    slices are new objects on every call).  A re-shuffled batch is a new combination: composed from the per-sample plans
    when the dicts carry them (their first use builds and caches them), else planned afresh over the union."""
    key = tuple(id(n) for n in nbr_dicts) + tuple((id(c), c._version) for c in parents)
    hit = _MERGE_CACHE.pop(key, None)          # (re-inserted below: the dict's order is the order of last use)
    if hit is None or any(a is not b for a, b in zip(hit[1], parents)) or any(a is not b for a, b in zip(hit[2], nbr_dicts)):
        # A union holds tens of MB of derived arrays (edge rows, statistics, orders).  Under a SHUFFLING loader no composition ever comes
        # back, and a cache that keeps the last 64 of them makes every step allocate fresh device memory until it is cleared (measured on
        # the NACA configuration: one ~120 MB hipMalloc every third step at ~75 ms each -- 8 ms steps became 40 ms on average,
        # tools/vx_shuffle_profile.py).  Keeping only the few most recently used lets the allocator hand the evicted union's blocks to the
        # next one; loaders that cycle through a handful of fixed batches still hit.
        # (a union a captured graph reads by raw pointer is pinned: never evicted, never released)
        for k in [k for k, v in _MERGE_CACHE.items() if not getattr(v[0], "pinned", False)][:max(0, len(_MERGE_CACHE) - _MERGE_CACHE_MAX + 1)]:
            _MERGE_CACHE.pop(k)[0].release()
        # a dict seen before (it carries a marker) is worth a plan of its own: the next batch that contains it composes
        seen = all(n.get("_gaot_amd_seen") for n in nbr_dicts)
        for n in nbr_dicts:
            n["_gaot_amd_seen"] = True
        hit = (MergedGeometry(nbr_dicts, src_coords, dst_coords, build_parts=seen), tuple(parents), list(nbr_dicts))   # hold refs: ids stay unique
    _MERGE_CACHE[key] = hit
    if torch.cuda.is_current_stream_capturing():
        hit[0].pinned = True          # the graph being captured holds this union's arrays by address (a rollout's per-step forward)
    return hit[0]


# ------------------------------------------------------------------------------------------------------------------------------
# Static padded unions: vx TRAINING at hipGraph-replay speed while the batch composition changes every step
# ------------------------------------------------------------------------------------------------------------------------------
VX_STATIC = os.environ.get("GAOT_VX_STATIC", "1") != "0"       # 0: the training path composes a fresh union per batch composition (rounds 3-5)
_UNION_SERIAL = itertools.count(1)


def edge_bucket(E: int) -> int:
    """edge capacity of the static union that serves a batch of E edges: E rounded up to 5 significant bits (steps of 3-6 %: a shuffling loader
    over one dataset lands in a handful of buckets, each with its own buffers and captured graphs; the pads cost the per-edge kernels that share)"""
    E = max(int(E), 1024)
    sh = max(E.bit_length() - 5, 0)
    return ((E + (1 << sh) - 1) >> sh) << sh


class _StaticPlan(GeometryPlan):
    """A plan whose arrays are STATIC buffers rewritten in place by device kernels (a StaticUnion re-composed from its table, a DropPlan re-drawn):
    launches a captured step replays.  Geometry-derived arrays (kernel-MLP rows, cosine weights, statistics, ...) are static buffers too, computed on
    REQUEST, once per version of the plan's arrays (`touch()` after every rewrite): in program order, so inside a capture they become graph nodes
    right where the eager pass would launch them.  The coordinates are whatever the caller passes at that moment (a union passes its own stacked
    buffers; a fixed mesh's coordinate tensors may be new objects every step)."""

    def __init__(self, Q: int, E: int, n_src: int, device):
        super().__init__(None, None, 0)
        self.Q, self.E, self.n_src = int(Q), int(E), int(n_src)
        dev = device
        self.index = torch.zeros(self.E, dtype=torch.int32, device=dev)
        self.edge_query = torch.zeros(self.E, dtype=torch.int32, device=dev)
        self._t_edge = torch.zeros(self.E, dtype=torch.int32, device=dev)
        self.splits = torch.zeros(self.Q + 1, dtype=torch.int32, device=dev)
        self._t_splits = torch.zeros(self.n_src + 1, dtype=torch.int32, device=dev)
        self._tcsr = None             # raw-composed unions: derives the transposed CSR from `index` on first request per version
        self._t_version = -1
        self.e_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.max_deg = self.max_t_deg = None          # unknown on the host: the edge-partitioned kernels serve any degree distribution
        self._src_id = None
        self._coord_cache, self._groups, self.epoch = {}, {}, 0
        self._version = 0
        self._arrays: Dict[str, list] = {}      # name -> [value, version it was computed at]

    def touch(self):
        """the plan's arrays were rewritten: every derived array is stale"""
        self._version += 1

    # the transposed CSR: written by whoever fills the plan (a union composed from per-sample plans, a DropPlan), or -- a union composed from raw
    # int64 lists -- derived from `index` the first time a kernel asks for it after a rewrite (the encoder's fused transform never does)
    def _need_t(self):
        if self._tcsr is not None and self._t_version != self._version:
            self._t_version = self._version
            self._tcsr()

    @property
    def t_splits(self):
        self._need_t()
        return self._t_splits

    @property
    def t_edge(self):
        self._need_t()
        return self._t_edge

    def _static(self, name, alloc, compute):
        ent = self._arrays.get(name)
        if ent is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError(f"gaot_amd: array {name!r} of a static plan requested for the first time inside a graph capture "
                                   "(the warm-up passes must run the same model path)")
            ent = self._arrays[name] = [alloc(), -1]
        if ent[1] != self._version:
            compute(ent[0])
            ent[1] = self._version
        return ent[0]

    # ---- overrides: no identity-keyed caches, no host-side sizes
    row_order = None

    @property
    def deg(self):
        return self._static("deg", lambda: torch.zeros(self.Q, dtype=torch.int64, device=self.splits.device),
                            lambda v: v.copy_(self.splits[1:] - self.splits[:-1]))

    @property
    def edge_query_long(self):
        return self._static("eq64", lambda: torch.zeros(self.E, dtype=torch.int64, device=self.splits.device), lambda v: v.copy_(self.edge_query))

    @property
    def index_long(self):
        return self._static("idx64", lambda: torch.zeros(self.E, dtype=torch.int64, device=self.splits.device), lambda v: v.copy_(self.index))

    @property
    def inv_deg_edge(self):
        def compute(v):
            L.check(L.load().gaot_edge_inv_degree(_p(self.splits), _p(self.edge_query), self.E, _p(self.e_dev), _p(v), _stream()), "gaot_edge_inv_degree")
        return self._static("inv_deg", lambda: torch.zeros(self.E, dtype=torch.float32, device=self.splits.device), compute)

    def edge_features(self, src, qry):
        s, q = src.contiguous(), qry.contiguous()

        def compute(feat):
            L.check(L.load().gaot_edge_features(_p(s), _p(q), s.shape[1], _p(self.index), _p(self.edge_query), self.E, _p(feat), None, _stream()),
                    "gaot_edge_features")
        return self._static(f"feat{s.shape[1]}", lambda: torch.zeros(self.E, 2 * s.shape[1], device=s.device, dtype=torch.float32), compute)

    def cosine_attention(self, src, qry):
        s, q = src.contiguous(), qry.contiguous()

        def compute(attn):
            lib = L.load()
            L.check(lib.gaot_edge_attention_cosine(_p(s), _p(q), s.shape[1], _p(self.index), _p(self.splits), self.Q, _p(attn), None, _stream()),
                    "gaot_edge_attention_cosine")
            L.check(lib.gaot_edge_zero_pads(_p(attn), 1, self.E, 1, _p(self.e_dev), self.E, _stream()), "gaot_edge_zero_pads")
        return self._static(f"cos{s.shape[1]}", lambda: torch.zeros(self.E, device=s.device, dtype=torch.float32), compute)

    def geo_stats(self, geom, qry, groups: int = 1):
        s, q = geom.contiguous(), qry.contiguous()
        F = 3 + 2 * s.shape[1]

        def alloc():
            return (torch.zeros(self.Q, F, device=s.device, dtype=torch.float32), torch.zeros(GEO_SCRATCH * F * groups, device=s.device, dtype=torch.float64))

        def compute(full):
            L.check(L.load().gaot_geo_stats(_p(s), _p(q), s.shape[1], _p(self.index), _p(self.splits), self.Q, _p(full[0]), _p(full[1]), None, groups,
                                            _stream()), "gaot_geo_stats")
        return self._static(f"stats{groups}", alloc, compute)[0]


DROP_RECORD: Optional[list] = None      # tests: a list here receives (index [kept] int64, row_splits [Q + 1] int64) on the HOST after every draw


class DropPlan(_StaticPlan):
    """Training-time neighbour sub-sampling on the device (reference edge_drop.py:54-99; MAGNOConfig.sampling_strategy 'ratio' / 'max_neighbors'):
    the sub-sampled graph of `base` -- any plan, also the padded union of a vx batch -- in static buffers of the base's capacity, re-drawn by
    `redraw()` with a device-resident seed (csrc/edge_drop.hip).  No host value depends on the draw, so a captured step draws a fresh subset on
    every replay; the kept edge count is `e_dev`, the rest of the buffers are pads (see StaticUnion)."""

    def __init__(self, base: GeometryPlan, strategy: str, max_neighbors, sample_ratio):
        super().__init__(base.Q, max(base.E, 1), base.n_src, base.splits.device)
        self.base = base
        self.mode = {"ratio": 1, "max_neighbors": 2}[strategy]
        self.max_neighbors = int(max_neighbors or 0)
        self.sample_ratio = float(sample_ratio or 0.0)
        dev = base.splits.device
        self._scratch = torch.zeros(int(L.load().gaot_edge_drop_scratch(self.E)), dtype=torch.int32, device=dev)
        self._seed = torch.zeros(1, dtype=torch.int64, device=dev)
        self.neighbors = {"neighbors_index": None, "neighbors_row_splits": None, _PLAN_KEY: self}

    def redraw(self) -> None:
        from . import ops
        lib = L.load()
        b = self.base
        if b.E == 0:
            self.splits.zero_(); self.t_splits.zero_(); self.e_dev.zero_()
            self.touch()
            return
        L.check(lib.gaot_attention_seed_next(_p(ops.dropout_state(self.splits.device)), 0x6564676564726f70, _p(self._seed), _stream()), "gaot_attention_seed_next")
        L.check(lib.gaot_edge_drop(_p(b.index), _p(b.edge_query), _p(b.t_edge), _p(b.splits), _p(b.t_splits), b.Q, b.n_src, b.E, _p(b.e_dev), self.mode,
                                   self.sample_ratio if self.mode == 1 else 1.0, self.max_neighbors if self.mode == 2 else 1, _p(self._seed),
                                   _p(self.index), _p(self.edge_query), _p(self.t_edge), _p(self.splits), _p(self.t_splits), _p(self.e_dev), _p(self._scratch),
                                   _stream()), "gaot_edge_drop")
        self.touch()
        if DROP_RECORD is not None and not torch.cuda.is_current_stream_capturing():
            e = int(self.e_dev.item())
            DROP_RECORD.append((self.index[:e].long().cpu(), self.splits.long().cpu()))


def dropped_plan(base: GeometryPlan, strategy: Optional[str], max_neighbors=None, sample_ratio=None) -> GeometryPlan:
    """`base` sub-sampled for this training pass (a fresh draw), or `base` itself where the reference returns the graph untouched: no strategy,
    'ratio' with sample_ratio None or >= 1, 'max_neighbors' without a limit (edge_drop.py:54-58, 74-76)"""
    if strategy == "ratio":
        if sample_ratio is None or sample_ratio >= 1.0:
            return base
        key = ("ratio", float(sample_ratio))
    elif strategy == "max_neighbors":
        if max_neighbors is None:
            return base
        key = ("max_neighbors", int(max_neighbors))
    else:
        return base
    drops = base.__dict__.setdefault("_drops", {})
    dp = drops.get(key)
    if dp is None:
        dp = drops[key] = DropPlan(base, strategy, max_neighbors, sample_ratio)
    dp.redraw()
    return dp


class StaticUnion:
    """Block-diagonal union of a vx batch in STATIC buffers padded to an edge capacity (`edge_bucket`), composed on the device from a small table
    of per-sample plan pointers (csrc/gno.hip union_compose_kernel).  What changes from step to step under a shuffling loader (the reference's
    default: data_utils.py:272-294, static_trainer.py:180-202) is only that table -- uploaded by `load()` from pinned memory, outside any graph --
    while `refresh()` (compose + the geometry-derived arrays: kernel-MLP rows, cosine weights, per-sample standardised statistics) is a fixed
    sequence of launches over fixed addresses: a captured training step begins with it and replays for ANY batch composition of the bucket.

    Edges past the real count are pads: no CSR row references them, the edge-partitioned kernels stop at the device-side count (`plan.e_dev`), the
    flat per-edge kernels (kernel MLP, edge gradients) do walk them and get an edge scale of exactly 0, so they add exact zeros."""

    RING = 8          # pinned staging slots of the table: a slot is reused only after the copy that read it has completed

    def __init__(self, B: int, n_src: int, n_dst: int, dim_src: int, dim_dst: int, e_cap: int, device):
        self.uid = next(_UNION_SERIAL)
        self.B, self.n_src, self.n_dst, self.e_cap, self.device = B, n_src, n_dst, int(e_cap), device
        self.n_src_each, self.n_dst_each = [n_src] * B, [n_dst] * B
        self.src = torch.zeros(B * n_src, dim_src, device=device, dtype=torch.float32)
        self.dst = torch.zeros(B * n_dst, dim_dst, device=device, dtype=torch.float32)
        self.plan = _StaticPlan(B * n_dst, int(e_cap), B * n_src, device)
        self.neighbors = {"neighbors_index": None, "neighbors_row_splits": None, _PLAN_KEY: self.plan}
        self.table = torch.zeros(B, 8, dtype=torch.int64, device=device)           # B x gaot_union_part (64 bytes each), or B x gaot_union_part_raw in its first 5 B words
        self.raw = False              # how the last load described the samples: per-sample plans, or the callers' raw int64 lists
        self.flag = torch.zeros(1, dtype=torch.int32, device=device)              # raw lists: CSR contract violations seen by the compose kernel
        self._tscratch = None
        self._since_check = 0
        self._pinned = [torch.zeros(B, 8, dtype=torch.int64).pin_memory() for _ in range(self.RING)]
        self._views = [t.numpy() for t in self._pinned]
        self._events: List[Optional[torch.cuda.Event]] = [None] * self.RING
        self._slot = 0
        self._hold = None
        self.pending = None
        self.pending_raw = False      # what the next load will be (set by whoever found this union for a batch): part of a captured step's identity
        self.e_real = 0           # host copy of the last loaded edge count (diagnostics; the kernels read plan.e_dev)

    def load(self, plans, src_parent: torch.Tensor, dst_parent: torch.Tensor) -> None:
        """upload the table of this batch: per-sample plan arrays (any order of any samples whose edge counts fit the capacity) and coordinates
        (`*_parent`: [B, n, d] per-sample or [n, d] shared by all samples, fp32 contiguous).  Not capturable; stream-ordered."""
        B = self.B
        if len(plans) != B:
            raise ValueError(f"static union of {B} samples handed {len(plans)}")
        counts = np.fromiter((p.E for p in plans), dtype=np.int64, count=B)
        begins = np.concatenate(([0], np.cumsum(counts)[:-1]))
        total = int(counts.sum())
        if total > self.e_cap:
            raise ValueError(f"batch of {total} edges does not fit the union's capacity {self.e_cap}")
        for p in plans:
            if p.Q != self.n_dst or p.n_src != self.n_src:
                raise ValueError("vx mode needs the same number of source / query points in every sample of a batch")
        slot = self._slot
        self._slot = (slot + 1) % self.RING
        ev = self._events[slot]
        if ev is not None:
            ev.synchronize()
        tab = self._views[slot]
        tab[:, :5] = [p.part_pointers() for p in plans]
        for col, parent, n in ((5, src_parent, self.n_src), (6, dst_parent, self.n_dst)):
            if parent.dtype != torch.float32 or not parent.is_contiguous() or parent.device != self.src.device:
                raise TypeError("static union: coordinates must be contiguous float32 tensors on the union's device")
            stride = n * parent.shape[-1] * 4 if parent.dim() == 3 else 0
            tab[:, col] = parent.data_ptr() + stride * np.arange(B, dtype=np.int64)
        tab[:, 7] = begins | (counts << 32)
        self.table.copy_(self._pinned[slot], non_blocking=True)
        if ev is None:
            ev = self._events[slot] = torch.cuda.Event()
        ev.record()
        self._hold = (list(plans), src_parent, dst_parent)          # the arrays behind the table's pointers live until the next load
        self.e_real = total
        self.raw = False

    CHECK_EVERY = 64      # raw lists: the device-side validity flag is read back (one host synchronisation) every so many loads

    def load_raw(self, dicts, src_parent: torch.Tensor, dst_parent: torch.Tensor) -> None:
        """as load(), from the callers' neighbour dicts themselves (int64 `neighbors_index` / `neighbors_row_splits` on the device): nothing is
        built per sample -- the compose kernel converts, offsets and validates, the transposed CSR is derived on the device when a kernel needs
        it.  For dicts that are new objects every step (the reference's trainer uploads them per step: move_to_device, static_trainer.py:192-193)."""
        B = self.B
        if len(dicts) != B:
            raise ValueError(f"static union of {B} samples handed {len(dicts)}")
        idx = [d["neighbors_index"] for d in dicts]
        sps = [d["neighbors_row_splits"] for d in dicts]
        for i_, s_ in zip(idx, sps):
            if i_.dtype != torch.int64 or s_.dtype != torch.int64 or not i_.is_contiguous() or not s_.is_contiguous() or i_.device != self.src.device:
                raise TypeError("static union: neighbour lists must be contiguous int64 tensors on the union's device")
            if s_.numel() != self.n_dst + 1:
                raise ValueError("vx mode needs the same number of source / query points in every sample of a batch")
        counts = np.fromiter((t.numel() for t in idx), dtype=np.int64, count=B)
        begins = np.concatenate(([0], np.cumsum(counts)[:-1]))
        total = int(counts.sum())
        if total > self.e_cap:
            raise ValueError(f"batch of {total} edges does not fit the union's capacity {self.e_cap}")
        self._since_check += 1
        if self._since_check >= self.CHECK_EVERY:
            self._since_check = 0
            _raise_if_bad(int(self.flag.item()))
        slot = self._slot
        self._slot = (slot + 1) % self.RING
        ev = self._events[slot]
        if ev is not None:
            ev.synchronize()
        tab = self._views[slot]
        tab[:, 0] = [t.data_ptr() for t in idx]
        tab[:, 1] = [t.data_ptr() for t in sps]
        for col, parent, n in ((2, src_parent, self.n_src), (3, dst_parent, self.n_dst)):
            if parent.dtype != torch.float32 or not parent.is_contiguous() or parent.device != self.src.device:
                raise TypeError("static union: coordinates must be contiguous float32 tensors on the union's device")
            stride = n * parent.shape[-1] * 4 if parent.dim() == 3 else 0
            tab[:, col] = parent.data_ptr() + stride * np.arange(B, dtype=np.int64)
        tab[:, 4] = begins | (counts << 32)
        self.table.copy_(self._pinned[slot], non_blocking=True)
        if ev is None:
            ev = self._events[slot] = torch.cuda.Event()
        ev.record()
        self._hold = (list(dicts), src_parent, dst_parent, idx, sps)
        self.e_real = total
        self.raw = True

    def _transpose(self):
        pl = self.plan
        if self._tscratch is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("gaot_amd: the transposed CSR of a raw-composed union requested for the first time inside a graph capture")
            self._tscratch = torch.zeros(int(L.load().gaot_csr_transpose_dev_scratch(pl.E, pl.n_src)), dtype=torch.int32, device=self.device)
        L.check(L.load().gaot_csr_transpose_dev(_p(pl.index), pl.E, _p(pl.e_dev), pl.n_src, _p(pl._t_splits), _p(pl._t_edge), _p(self._tscratch), _stream()),
                "gaot_csr_transpose_dev")

    def load_pending(self, src_parent: torch.Tensor, dst_parent: torch.Tensor) -> None:
        """load() with the plans a `vx_unions(..., load=False)` call left behind"""
        plans, self.pending = self.pending, None
        if plans is None:           # loaded already for this batch (the call that captured an entry replays it at once): the same samples again
            plans = self._hold[0]
        if isinstance(plans[0], dict):
            self.load_raw(plans, src_parent, dst_parent)
        else:
            self.load(plans, src_parent, dst_parent)

    def refresh(self) -> None:
        """compose the union from the loaded table (capturable: a fixed launch over fixed addresses); the derived arrays follow on request"""
        if self._hold is None:
            raise RuntimeError("gaot_amd: StaticUnion.refresh() before any load(): the device table holds no sample pointers yet")
        pl = self.plan
        if self.raw:
            L.check(L.load().gaot_union_compose_raw(_p(self.table), self.B, self.n_dst, self.n_src, self.src.shape[1], self.dst.shape[1], self.e_cap,
                                                    _p(pl.index), _p(pl.edge_query), _p(pl.splits), _p(self.src), _p(self.dst), _p(pl.e_dev), _p(self.flag),
                                                    _stream()), "gaot_union_compose_raw")
            pl._tcsr = self._transpose
        else:
            pl._tcsr = None
            L.check(L.load().gaot_union_compose(_p(self.table), self.B, self.n_dst, self.n_src, self.src.shape[1], self.dst.shape[1], self.e_cap,
                                                _p(pl.index), _p(pl.edge_query), _p(pl._t_edge), _p(pl.splits), _p(pl._t_splits), _p(self.src), _p(self.dst),
                                                _p(pl.e_dev), _stream()), "gaot_union_compose")
        pl.touch()

    def geo_stats(self) -> torch.Tensor:
        return self.plan.geo_stats(self.src, self.dst, groups=self.B)
