"""CPU oracle for the GAOT forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``gaot_amd/`` imports this file.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import it, and there only as the checker / the timed CPU baseline.

It is a from-scratch restatement, in plain fp32 PyTorch on the CPU (no torch_scatter,
no torch_cluster), of the algorithm in camlab-ethz/GAOT ``src/model`` (every function
cites the reference file:line it follows).  It is *functional*: weights come in as a
``state_dict``-style mapping that uses the reference's parameter names, so golden
vectors exported from the imported reference plug in directly.

Pinning: ``tests/golden/*.npz`` were produced in the build container by importing the
reference itself (``tests/golden/make_golden.py``; the absent third-party
``torch_scatter`` was replaced by a pure-torch stand-in with upstream semantics:
empty segment -> 0, mean divides by max(count, 1)).  ``tests/test_oracle_golden.py``
checks this oracle against every one of those vectors.  What stays unpinned is only
the behaviour of the real ``torch_scatter``/``torch_cluster`` binaries, which the
reference does not vendor or version-pin.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
CSR = Tuple[Tensor, Tensor]  # (neighbors_index [E] int64, neighbors_row_splits [Q+1] int64)


# --------------------------------------------------------------------------------------
# configuration (field names follow MAGNOConfig magno.py:31-60 and
# TransformerConfig/AttentionConfig attn.py:21-38; flattened into one record)
# --------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    # MAGNO
    coord_dim: int = 2
    radius: float = 0.033
    hidden_size: int = 64
    mlp_layers: int = 3
    lifting_channels: int = 32
    scales: List[float] = field(default_factory=lambda: [1.0])
    use_scale_weights: bool = False
    use_attention: bool = True
    attention_type: str = "cosine"
    use_geoembed: bool = True
    embedding_method: str = "statistical"
    pooling: str = "max"
    transform_type: str = "linear"
    node_embedding: bool = False
    precompute_edges: bool = False
    # transformer
    patch_size: int = 8
    tf_hidden_size: int = 256
    use_attn_norm: bool = True
    use_ffn_norm: bool = True
    norm_eps: float = 1e-6
    num_layers: int = 3
    positional_embedding: str = "absolute"
    use_long_range_skip: bool = True
    ffn_multiplier: int = 4
    num_heads: int = 8
    num_kv_heads: int = 8
    use_conditional_norm: bool = False
    # model
    latent_tokens_size: List[int] = field(default_factory=lambda: [64, 64])
    # test instrument, not a reference option: evaluate the geometry statistics (gemb.py:103-171) in this dtype before the
    # embedding MLP.  "float64" separates the CONDITIONING of a result w.r.t. the reference's fp32 rounding of those
    # statistics (ReLU gates of geoembed.mlp.0 sit right behind them) from an implementation error.
    stats_dtype: str = "float32"
    # baseline option (bench.py's torch-on-the-GPU leg): call F.scaled_dot_product_attention, the library op the reference itself
    # calls (attn.py:114), instead of the written-out softmax -- the same function up to rounding (tests/test_oracle_golden.py)
    library_attention: bool = False


def as_csr(n) -> CSR:
    """Accept the reference's dict form (neighbor_search.py:139-140) or a tuple."""
    if isinstance(n, dict):
        return n["neighbors_index"], n["neighbors_row_splits"]
    return n


# --------------------------------------------------------------------------------------
# radius graph -- restates _native_neighbor_search (neighbor_search.py:108-146):
# inclusive `dist <= r`, unbounded degree, neighbours in ascending data index.
# --------------------------------------------------------------------------------------
def radius_csr(data: Tensor, queries: Tensor, radius: float, chunk: int = 4096, exact: bool = False) -> CSR:
    """exact=False: the `native` backend's test, torch.cdist(...) <= r (neighbor_search.py:108-146; cdist expands
    |q|^2 + |d|^2 - 2 q.d for large inputs, so pairs within ~1e-6 of r can fall either side).
    exact=True: the distance test of the `grid` backend -- what method='auto' resolves to without torch_cluster --
    torch.norm(query - data[j]) <= r on explicit differences (neighbor_search.py:250-253); neighbours are returned in
    ascending data index here (the grid backend lists them cell by cell: same sets, different order inside a row)."""
    r = torch.tensor(radius, dtype=queries.dtype)
    cols, counts = [], []
    if exact:
        chunk = max(1, min(chunk, (32 << 20) // max(1, data.shape[0] * data.shape[1])))
    for s in range(0, queries.shape[0], chunk):
        if exact:
            d = torch.linalg.vector_norm(queries[s:s + chunk, None, :] - data[None, :, :], dim=-1)
        else:
            d = torch.cdist(queries[s:s + chunk], data)
        hit = d <= r
        cols.append(hit.nonzero()[:, 1])
        counts.append(hit.sum(dim=1))
    idx = torch.cat(cols).long()
    splits = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(torch.cat(counts), 0)]).long()
    return idx, splits


def radius_csr_torch_cluster(data: Tensor, queries: Tensor, radius: float, max_num_neighbors: int = 32) -> CSR:
    """What `_torch_cluster_neighbor_search` (neighbor_search.py:148-175) gets from `torch_cluster.radius(data, queries, r)`.
    torch_cluster is NOT vendored or version-pinned by the reference and is absent here: PARITY UNPINNED for this function.
    Restated from the published CUDA kernel (rusty1s/pytorch_cluster, csrc/cuda/radius_cuda.cu, 1.6.x): one thread per
    query scans the data points in index order, keeps a point when the squared distance is STRICTLY below r*r and stops
    once max_num_neighbors (default 32) are found.  The reference then turns (row, col) into CSR with bincount/cumsum."""
    r2 = torch.tensor(radius, dtype=queries.dtype) ** 2
    idx, counts = [], []
    for q in range(queries.shape[0]):                 # a literal restatement: small cases only
        d2 = ((data - queries[q]) ** 2).sum(-1)
        hit = (d2 < r2).nonzero()[:, 0][:max_num_neighbors]
        idx.append(hit)
        counts.append(hit.numel())
    index = torch.cat(idx).long() if idx else torch.zeros(0, dtype=torch.long)
    splits = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(torch.tensor(counts, dtype=torch.long), 0)])
    return index, splits


def latent_grid(sizes: Sequence[int], lo: float = -1.0, hi: float = 1.0) -> Tensor:
    """meshgrid(linspace)... 'ij' then flattened (data_processor.py:289-294)."""
    axes = [torch.linspace(lo, hi, n) for n in sizes]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, len(sizes))


# --------------------------------------------------------------------------------------
# CSR segment helpers (semantics of torch_scatter.segment_csr as used at
# agno.py:131-141,271: empty segment -> 0)
# --------------------------------------------------------------------------------------
def edge_query_ids(splits: Tensor) -> Tuple[Tensor, Tensor]:
    deg = splits[1:] - splits[:-1]
    return torch.repeat_interleave(torch.arange(deg.numel()), deg), deg


def seg_sum(src: Tensor, qid: Tensor, Q: int) -> Tensor:
    """src [..., E, *] summed per segment along dim -2 (or dim 0 for 1-D)."""
    if src.dim() == 1:
        return torch.zeros(Q, dtype=src.dtype).index_add_(0, qid, src)
    dim = src.dim() - 2
    shape = list(src.shape)
    shape[dim] = Q
    return torch.zeros(shape, dtype=src.dtype).index_add_(dim, qid, src)


def seg_max(src: Tensor, qid: Tensor, Q: int) -> Tensor:
    out = torch.zeros(Q, dtype=src.dtype)
    return out.scatter_reduce(0, qid, src, reduce="amax", include_self=False)


def segment_softmax(scores: Tensor, qid: Tensor, Q: int) -> Tensor:
    """agno.py:112-146."""
    m = seg_max(scores, qid, Q)
    ex = torch.exp(scores - m[qid])
    den = seg_sum(ex, qid, Q)
    return ex / den[qid]


# --------------------------------------------------------------------------------------
# small layers
# --------------------------------------------------------------------------------------
def gelu_mlp(sd: Dict[str, Tensor], prefix: str, x: Tensor) -> Tensor:
    """LinearChannelMLP (mlp.py:307-337): Linear + exact-erf GELU, last layer bare."""
    n = 0
    while f"{prefix}.fcs.{n}.weight" in sd:
        n += 1
    for i in range(n):
        x = x @ sd[f"{prefix}.fcs.{i}.weight"].t() + sd[f"{prefix}.fcs.{i}.bias"]
        if i < n - 1:
            x = F.gelu(x)
    return x


def pointwise_conv(sd: Dict[str, Tensor], prefix: str, x: Tensor) -> Tensor:
    """ChannelMLP with n_layers=1 (mlp.py:274-275,295): one Conv1d(k=1), no activation.
    x is channels-last [B, n, cin]; the reference permutes around the conv
    (magno.py:273-274,349-350,640-641)."""
    w = sd[f"{prefix}.fcs.0.weight"][:, :, 0]
    return x @ w.t() + sd[f"{prefix}.fcs.0.bias"]


def node_pos_encode(x: Tensor, freq: int = 4) -> Tensor:
    """gemb.py:12-34 -> [n, freq*2*d] ordered (freq, [sin d..., cos d...])."""
    fr = torch.arange(1, freq + 1).to(x.dtype)
    ang = fr[None, :, None] * (math.pi * (x + 1))[:, None, :]
    return torch.cat([ang.sin(), ang.cos()], dim=2).reshape(x.shape[0], -1)


# --------------------------------------------------------------------------------------
# AGNO integral transform (agno.py:148-273)
# --------------------------------------------------------------------------------------
def agno(sd, prefix: str, cfg: OracleConfig, y: Tensor, x: Tensor, f_y: Tensor, nbrs: CSR,
         rec: Optional[dict] = None, tag: str = "") -> Tensor:
    idx, splits = as_csr(nbrs)
    Q = splits.numel() - 1
    qid, deg = edge_query_ids(splits)
    yj = y[idx]                                   # neighbour (source) coords   agno.py:188
    xi = x[qid]                                   # query coords per edge        agno.py:206-207
    B = f_y.shape[0]
    fj = f_y[:, idx, :]                           # [B,E,C]                      agno.py:198

    att = None
    if cfg.use_attention:
        if cfg.attention_type == "cosine":        # agno.py:218-221 (F.normalize eps=1e-12)
            xn = xi / xi.norm(dim=-1, keepdim=True).clamp_min(1e-12)
            yn = yj / yj.norm(dim=-1, keepdim=True).clamp_min(1e-12)
            s = (xn * yn).sum(-1)
        elif cfg.attention_type == "dot_product":  # agno.py:106-110,215-217
            qv = xi @ sd[f"{prefix}.query_proj.weight"].t() + sd[f"{prefix}.query_proj.bias"]
            kv = yj @ sd[f"{prefix}.key_proj.weight"].t() + sd[f"{prefix}.key_proj.bias"]
            s = (qv * kv).sum(-1) * (1.0 / math.sqrt(64.0))
        else:
            raise ValueError(cfg.attention_type)
        att = segment_softmax(s, qid, Q)

    feat = torch.cat([yj, xi], dim=-1)            # order (y_j, x_i)             agno.py:229
    nonlinear = cfg.transform_type in ("nonlinear", "nonlinear_kernelonly")
    if nonlinear:                                 # agno.py:230-239
        feat = torch.cat([feat.unsqueeze(0).expand(B, -1, -1), fj], dim=-1)
    k = gelu_mlp(sd, f"{prefix}.channel_mlp", feat)   # [E,C] or [B,E,C]         agno.py:242
    if rec is not None:
        rec[f"{tag}kernel"] = k.detach()
        if att is not None:
            rec[f"{tag}attn"] = att.detach()
    if cfg.transform_type != "nonlinear_kernelonly":  # agno.py:245-246
        k = k * fj
    elif k.dim() == 2:
        k = k.unsqueeze(0).expand(B, -1, -1)
    if att is not None:                           # agno.py:249-250
        k = k * att[None, :, None]
    out = seg_sum(k, qid, Q)                      # agno.py:262-271
    if att is None:                               # 'mean' when no attention
        out = out / deg.clamp(min=1).to(out.dtype)[None, :, None]
    return out


# --------------------------------------------------------------------------------------
# geometric embedding (gemb.py:83-171 statistical, 173-228 pointnet)
# --------------------------------------------------------------------------------------
def geo_stats_raw(geom: Tensor, queries: Tensor, nbrs: CSR) -> Tensor:
    """Un-normalised [Q, 3+2d]: N_i, mean dist, var dist, centroid-query, eig(cov) desc."""
    idx, splits = as_csr(nbrs)
    Q, d = queries.shape
    qid, deg = edge_query_ids(splits)
    cnt = deg.to(geom.dtype)
    has = cnt > 0
    safe = cnt.clamp(min=1)
    nb = geom[idx]
    dist = (nb - queries[qid]).norm(dim=1)                        # gemb.py:118
    mean_d = seg_sum(dist, qid, Q) / safe                         # gemb.py:123
    mean_d2 = seg_sum(dist * dist, qid, Q) / safe                 # gemb.py:126-127
    var_d = (mean_d2 - mean_d * mean_d).clamp(min=0.0)            # gemb.py:128-131
    cen = seg_sum(nb, qid, Q) / safe[:, None]                     # gemb.py:134
    delta = cen - queries                                         # gemb.py:135
    ctr = nb - cen[qid]                                           # gemb.py:138
    cov = seg_sum((ctr[:, :, None] * ctr[:, None, :]).reshape(-1, d * d), qid, Q)
    cov = (cov / safe[:, None]).reshape(Q, d, d)                  # gemb.py:139-143
    pca = torch.zeros(Q, d, dtype=geom.dtype)
    if has.any():                                                 # gemb.py:146-153
        pca[has] = torch.linalg.eigvalsh(cov[has]).flip(dims=[1])
    feats = torch.cat([cnt[:, None], mean_d[:, None], var_d[:, None], delta, pca], dim=1)
    feats[~has] = 0.0                                             # gemb.py:161
    return feats


def geo_stats(geom: Tensor, queries: Tensor, nbrs: CSR) -> Tensor:
    """Global standardisation over queries, unbiased std, std<1e-6 -> 1 (gemb.py:164-169)."""
    raw = geo_stats_raw(geom, queries, nbrs)
    mu = raw.mean(dim=0, keepdim=True)
    sg = raw.std(dim=0, keepdim=True)
    sg = torch.where(sg < 1e-6, torch.ones_like(sg), sg)
    return (raw - mu) / sg


def geoembed(sd, prefix: str, cfg: OracleConfig, geom: Tensor, queries: Tensor, nbrs: CSR,
             rec: Optional[dict] = None, tag: str = "") -> Tensor:
    if cfg.embedding_method == "statistical":                      # gemb.py:54-59,230-233
        if cfg.stats_dtype == "float64":
            st = geo_stats(geom.double(), queries.double(), nbrs).to(geom.dtype)
        else:
            st = geo_stats(geom, queries, nbrs)
        if rec is not None:
            rec[f"{tag}geo_stats"] = st.detach()
        h = torch.relu(st @ sd[f"{prefix}.mlp.0.weight"].t() + sd[f"{prefix}.mlp.0.bias"])
        return torch.relu(h @ sd[f"{prefix}.mlp.2.weight"].t() + sd[f"{prefix}.mlp.2.bias"])
    # pointnet variant (gemb.py:173-228)
    idx, splits = as_csr(nbrs)
    Q = queries.shape[0]
    qid, deg = edge_query_ids(splits)
    out_dim = sd[f"{prefix}.fc.0.weight"].shape[0]
    res = torch.zeros(Q, out_dim, dtype=geom.dtype)
    has = deg > 0
    if not has.any():
        return res
    rel = geom[idx] - queries[qid]
    h = torch.relu(rel @ sd[f"{prefix}.pointnet_mlp.0.weight"].t() + sd[f"{prefix}.pointnet_mlp.0.bias"])
    h = torch.relu(h @ sd[f"{prefix}.pointnet_mlp.2.weight"].t() + sd[f"{prefix}.pointnet_mlp.2.bias"])
    if cfg.pooling == "max":
        pooled = torch.zeros(Q, h.shape[1], dtype=h.dtype).scatter_reduce(
            0, qid[:, None].expand_as(h), h, reduce="amax", include_self=False)
    else:
        pooled = seg_sum(h, qid, Q) / deg.clamp(min=1).to(h.dtype)[:, None]
    emb = torch.relu(pooled @ sd[f"{prefix}.fc.0.weight"].t() + sd[f"{prefix}.fc.0.bias"])
    return torch.where(has[:, None], emb, res)


# --------------------------------------------------------------------------------------
# MAGNO encoder / decoder (magno.py:217-413, 552-751)
# --------------------------------------------------------------------------------------
def _kcoord(cfg: OracleConfig, c: Tensor) -> Tensor:
    return node_pos_encode(c) if cfg.node_embedding else c


def _scale_weights(sd, prefix: str, coords: Tensor) -> Tensor:
    h = torch.relu(coords @ sd[f"{prefix}.scale_weighting.0.weight"].t() + sd[f"{prefix}.scale_weighting.0.bias"])
    return torch.softmax(h @ sd[f"{prefix}.scale_weighting.2.weight"].t() + sd[f"{prefix}.scale_weighting.2.bias"], dim=-1)


def _combine_scales(cfg, per_scale: List[Tensor], w: Optional[Tensor]) -> Tensor:
    if len(per_scale) == 1:
        return per_scale[0]
    if cfg.use_scale_weights:                                      # magno.py:295-300
        acc = torch.zeros_like(per_scale[0])
        for i, t in enumerate(per_scale):
            acc = acc + w[None, :, i:i + 1] * t
        return acc
    return torch.stack(per_scale, 0).mean(0)                       # magno.py:303


def _one_transform(sd, side: str, cfg, src_coord, dst_coord, feats, nbrs, rec, tag):
    """AGNO (+ geoembed + recovery) for ONE geometry and ONE scale.
    src = points integrated over (y), dst = query points (x)."""
    out = agno(sd, f"{side}.agno", cfg, _kcoord(cfg, src_coord), _kcoord(cfg, dst_coord), feats, nbrs, rec, tag)
    if rec is not None:
        rec[f"{tag}agno"] = out.detach()
    if cfg.use_geoembed:
        ge = geoembed(sd, f"{side}.geoembed", cfg, src_coord, dst_coord, nbrs, rec, tag)
        if rec is not None:
            rec[f"{tag}geoembed"] = ge.detach()
        cat = torch.cat([out, ge.unsqueeze(0).expand(out.shape[0], -1, -1)], dim=-1)
        out = pointwise_conv(sd, f"{side}.recovery", cat)
    return out


def _neighbors(cfg, src, dst, given, fx: bool):
    """Either the caller's lists (precompute_edges) or a radius search per scale
    (magno.py:174-215,510-550).  fx: List[scale]; vx: List[batch][scale]."""
    if given is not None:
        return given
    if fx:
        return [radius_csr(src, dst, cfg.radius * s) for s in cfg.scales]
    B = max(src.shape[0] if src.dim() == 3 else 0, dst.shape[0] if dst.dim() == 3 else 0)
    return [[radius_csr(src[b] if src.dim() == 3 else src, dst[b] if dst.dim() == 3 else dst, cfg.radius * s)
             for s in cfg.scales] for b in range(B)]


def magno_encode(sd, cfg: OracleConfig, x_coord: Tensor, pndata: Tensor, latent: Tensor,
                 nbrs=None, rec: Optional[dict] = None) -> Tensor:
    fx = x_coord.dim() == 2
    nb = _neighbors(cfg, x_coord, latent, nbrs, fx)
    lifted = pointwise_conv(sd, "encoder.lifting", pndata)         # magno.py:273-274
    if rec is not None:
        rec["enc.lifted"] = lifted.detach()
    w = _scale_weights(sd, "encoder", _kcoord(cfg, latent)) if cfg.use_scale_weights else None
    per_scale = []
    for si in range(len(cfg.scales)):
        tag = f"enc.s{si}."
        if fx:                                                     # magno.py:307-354
            per_scale.append(_one_transform(sd, "encoder", cfg, x_coord, latent, lifted, nb[si], rec, tag))
        else:                                                      # magno.py:356-413
            per_scale.append(torch.cat([
                _one_transform(sd, "encoder", cfg, x_coord[b], latent, lifted[b:b + 1], nb[b][si], None, tag)
                for b in range(x_coord.shape[0])], dim=0))
    out = _combine_scales(cfg, per_scale, w)
    if rec is not None:
        rec["enc.out"] = out.detach()
    return out


def magno_decode(sd, cfg: OracleConfig, latent: Tensor, rndata: Tensor, query: Tensor,
                 nbrs=None, rec: Optional[dict] = None) -> Tensor:
    fx = query.dim() == 2
    nb = _neighbors(cfg, latent, query, nbrs, fx)
    w = None
    if cfg.use_scale_weights:                                      # magno.py:607-613 (vx: sample 0)
        w = _scale_weights(sd, "decoder", _kcoord(cfg, query if fx else query[0]))
    per_scale = []
    for si in range(len(cfg.scales)):
        tag = f"dec.s{si}."
        if fx:                                                     # magno.py:645-692
            per_scale.append(_one_transform(sd, "decoder", cfg, latent, query, rndata, nb[si], rec, tag))
        else:                                                      # magno.py:694-751
            per_scale.append(torch.cat([
                _one_transform(sd, "decoder", cfg, latent, query[b], rndata[b:b + 1], nb[b][si], None, tag)
                for b in range(query.shape[0])], dim=0))
    dec = _combine_scales(cfg, per_scale, w)
    if rec is not None:
        rec["dec.pre_projection"] = dec.detach()
    return pointwise_conv(sd, "decoder.projection", dec)           # magno.py:640-641


# --------------------------------------------------------------------------------------
# processor (gaot.py:145-233, attn.py)
# --------------------------------------------------------------------------------------
def patchify(x: Tensor, sizes: Sequence[int], P: int) -> Tensor:
    """[B, prod(sizes), C] -> [B, S, P^d C]   (gaot.py:182-185, 202-205)."""
    B, _, C = x.shape
    if len(sizes) == 2:
        H, W = sizes
        t = x.reshape(B, H // P, P, W // P, P, C).permute(0, 1, 3, 2, 4, 5)
        return t.reshape(B, (H // P) * (W // P), P * P * C)
    H, W, D = sizes
    t = x.reshape(B, H // P, P, W // P, P, D // P, P, C).permute(0, 1, 3, 5, 2, 4, 6, 7)
    return t.reshape(B, (H // P) * (W // P) * (D // P), P * P * P * C)


def unpatchify(x: Tensor, sizes: Sequence[int], P: int, C: int) -> Tensor:
    """inverse (gaot.py:224-231)."""
    B = x.shape[0]
    if len(sizes) == 2:
        H, W = sizes
        t = x.reshape(B, H // P, W // P, P, P, C).permute(0, 1, 3, 2, 4, 5)
        return t.reshape(B, H * W, C)
    H, W, D = sizes
    t = x.reshape(B, H // P, W // P, D // P, P, P, P, C).permute(0, 1, 4, 2, 5, 3, 6, 7)
    return t.reshape(B, H * W * D, C)


def patch_positions(sizes: Sequence[int], P: int) -> Tensor:
    axes = [torch.arange(n // P, dtype=torch.float32) for n in sizes]   # gaot.py:92-117
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, len(sizes))


def absolute_pos_embedding(pos: Tensor, embed_dim: int) -> Tensor:
    """gaot.py:119-130: per axis [sin(f_0..f_{n-1}), cos(...)], axes concatenated."""
    nd = pos.shape[1]
    n = embed_dim // (2 * nd)
    inv = 1.0 / (10000 ** (torch.arange(n, dtype=torch.float32) / n))
    ang = pos[:, :, None] * inv[None, None, :]
    return torch.cat([ang.sin(), ang.cos()], dim=-1).reshape(pos.shape[0], -1)


def rms_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w   # attn.py:167-172


def cond_norm(sd, prefix: str, c: Tensor, x: Tensor) -> Tensor:
    """ConditionedNorm with num_layers=2 -> one Linear each (mlp.py:49-52,108-124)."""
    sc = 1 + c * (c @ sd[f"{prefix}.mlp_scale.layers.0.weight"].t() + sd[f"{prefix}.mlp_scale.layers.0.bias"])
    bi = c * (c @ sd[f"{prefix}.mlp_bias.layers.0.weight"].t() + sd[f"{prefix}.mlp_bias.layers.0.bias"])
    return x * sc[:, None, :] + bi[:, None, :]


def rotate_queries_or_keys(t: Tensor, freqs: Tensor) -> Tensor:
    """`RotaryEmbedding(dim=head_dim).rotate_queries_or_keys(t)` as called at attn.py:106-108 on [B, heads, S, head_dim].
    The package (lucidrains/rotary-embedding-torch) is a runtime dependency the reference neither vendors nor pins and it is
    absent here: PARITY UNPINNED for this function; restated from its published source --
      freqs = 1 / theta^(arange(0, dim, 2) / dim), theta = 10000       (freqs_for='lang'; a non-trainable nn.Parameter)
      angle[s, 2i] = angle[s, 2i+1] = s * freqs[i]                      (positions = arange(seq_len); repeat '... n -> ... (n r)', r=2)
      out = t * cos(angle) + rotate_half(t) * sin(angle),   rotate_half: (x_{2i}, x_{2i+1}) -> (-x_{2i+1}, x_{2i})."""
    S = t.shape[-2]
    ang = torch.arange(S, dtype=t.dtype)[:, None] * freqs[None, :]
    ang = ang.repeat_interleave(2, dim=-1)
    x = t.reshape(*t.shape[:-1], -1, 2)
    rot = torch.stack((-x[..., 1], x[..., 0]), dim=-1).reshape(t.shape)
    return t * ang.cos() + rot * ang.sin()


def attention(sd, prefix: str, cfg: OracleConfig, x: Tensor, condition, drop_factor: Optional[Tensor] = None) -> Tensor:
    """GroupQueryFlashAttention.forward (attn.py:78-119), softmax attention written out.
    drop_factor [B, H, S, S] (optional): the dropout multiplier keep / (1 - p) applied to the softmax output, i.e. what
    F.scaled_dot_product_attention(dropout_p=p) does in training (attn.py:110-114) for ONE given draw of the mask."""
    if cfg.use_conditional_norm:
        x = cond_norm(sd, f"{prefix}.correction", condition, x)
    B, S, _ = x.shape
    H, Hkv = cfg.num_heads, cfg.num_kv_heads
    q = x @ sd[f"{prefix}.q_proj.weight"].t()
    k = x @ sd[f"{prefix}.k_proj.weight"].t()
    v = x @ sd[f"{prefix}.v_proj.weight"].t()
    dh = q.shape[-1] // H
    q = q.reshape(B, S, H, dh).transpose(1, 2)
    k = k.reshape(B, S, Hkv, dh).transpose(1, 2)
    v = v.reshape(B, S, Hkv, dh).transpose(1, 2)
    if Hkv != H:
        k = k.repeat_interleave(H // Hkv, dim=1)
        v = v.repeat_interleave(H // Hkv, dim=1)
    if cfg.positional_embedding == "rope":                              # attn.py:106-108
        q = rotate_queries_or_keys(q, sd[f"{prefix}.rotary_emb.freqs"])
        k = rotate_queries_or_keys(k, sd[f"{prefix}.rotary_emb.freqs"])
    if cfg.library_attention and drop_factor is None:                   # attn.py:114 as the reference calls it (no dropout draw to inject)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0).transpose(1, 2).reshape(B, S, H * dh)
        return o @ sd[f"{prefix}.o_proj.weight"].t()
    p = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
    if drop_factor is not None:
        p = p * drop_factor
    o = (p @ v).transpose(1, 2).reshape(B, S, H * dh)
    return o @ sd[f"{prefix}.o_proj.weight"].t()


def ffn(sd, prefix: str, cfg: OracleConfig, x: Tensor, condition) -> Tensor:
    g = F.silu(x @ sd[f"{prefix}.w1.weight"].t()) * (x @ sd[f"{prefix}.w3.weight"].t())
    y = g @ sd[f"{prefix}.w2.weight"].t()                              # attn.py:150-156
    if cfg.use_conditional_norm:
        y = cond_norm(sd, f"{prefix}.correction", condition, y)
    return y


def transformer_block(sd, prefix: str, cfg: OracleConfig, x: Tensor, condition, skip=None) -> Tensor:
    if skip is not None:                                               # attn.py:225-227
        x = torch.cat([x, skip], dim=-1) @ sd[f"{prefix}.skip_proj.weight"].t() + sd[f"{prefix}.skip_proj.bias"]
    h = rms_norm(x, sd[f"{prefix}.attn_norm.weight"], cfg.norm_eps) if cfg.use_attn_norm else x
    h = x + attention(sd, f"{prefix}.attn", cfg, h, condition)         # attn.py:229-230
    h = rms_norm(h, sd[f"{prefix}.ffn_norm.weight"], cfg.norm_eps) if cfg.use_ffn_norm else h
    return h + ffn(sd, f"{prefix}.ffn", cfg, h, condition)             # attn.py:231-232 (residual on normed h)


def transformer(sd, cfg: OracleConfig, x: Tensor, condition, rec: Optional[dict] = None) -> Tensor:
    if "processor.input_proj.weight" in sd:                            # attn.py:250-260
        x = x @ sd["processor.input_proj.weight"].t() + sd["processor.input_proj.bias"]
    half = cfg.num_layers // 2
    skips = []
    for i in range(half):
        x = transformer_block(sd, f"processor.encoder_layers.{i}", cfg, x, condition)
        skips.append(x)
        if rec is not None:
            rec[f"proc.enc{i}"] = x.detach()
    if cfg.num_layers % 2 == 1:
        x = transformer_block(sd, "processor.middle_layer", cfg, x, condition)
        if rec is not None:
            rec["proc.mid"] = x.detach()
    for i in range(half):
        skip = skips.pop() if cfg.use_long_range_skip else None
        x = transformer_block(sd, f"processor.decoder_layers.{i}", cfg, x, condition, skip=skip)
        if rec is not None:
            rec[f"proc.dec{i}"] = x.detach()
    if "processor.output_proj.weight" in sd:
        x = x @ sd["processor.output_proj.weight"].t() + sd["processor.output_proj.bias"]
    return x


def process(sd, cfg: OracleConfig, rndata: Tensor, condition=None, rec: Optional[dict] = None) -> Tensor:
    P, C = cfg.patch_size, rndata.shape[2]
    tok = patchify(rndata, cfg.latent_tokens_size, P)
    tok = tok @ sd["patch_linear.weight"].t() + sd["patch_linear.bias"]   # gaot.py:208
    if cfg.positional_embedding == "absolute":                            # gaot.py:212-216
        tok = tok + absolute_pos_embedding(patch_positions(cfg.latent_tokens_size, P), tok.shape[-1])
    elif cfg.positional_embedding != "rope":                              # gaot.py:217-218: rope acts inside attention
        raise ValueError(cfg.positional_embedding)
    if rec is not None:
        rec["proc.tokens"] = tok.detach()
    tok = transformer(sd, cfg, tok, condition, rec)
    return unpatchify(tok, cfg.latent_tokens_size, P, C)


# --------------------------------------------------------------------------------------
# GAOT.forward (gaot.py:248-305) and the trainer step (static_trainer.py:160-178,
# optimizers.py:247-257, loss base_trainer.py:71)
# --------------------------------------------------------------------------------------
def gaot_forward(sd, cfg: OracleConfig, latent: Tensor, xcoord: Tensor, pndata: Tensor,
                 query_coord: Optional[Tensor] = None, encoder_nbrs=None, decoder_nbrs=None,
                 condition: Optional[Tensor] = None, rec: Optional[dict] = None) -> Tensor:
    rn = magno_encode(sd, cfg, xcoord, pndata, latent, encoder_nbrs, rec)
    rn = process(sd, cfg, rn, condition, rec)
    if rec is not None:
        rec["proc.out"] = rn.detach()
    q = xcoord if query_coord is None else query_coord
    return magno_decode(sd, cfg, latent, rn, q, decoder_nbrs, rec)


def adamw_update(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float,
                 weight_decay: float, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
    """torch.optim.AdamW semantics (decoupled decay), the optimizer at optimizers.py:196."""
    p = p * (1 - lr * weight_decay)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mhat = m / (1 - b1 ** step)
    vhat = v / (1 - b2 ** step)
    return p - lr * mhat / (vhat.sqrt() + eps), m, v


def train_step(sd: Dict[str, Tensor], cfg: OracleConfig, batch: dict, lr: float = 8e-4,
               weight_decay: float = 1e-5, state: Optional[dict] = None, return_pred: bool = False):
    """One step: zero_grad -> forward -> MSE(mean) -> backward -> AdamW.
    Returns (loss, grads, new_sd, state) (+ the prediction when return_pred)."""
    params = {k: v.detach().clone().requires_grad_(not k.endswith("rotary_emb.freqs")) for k, v in sd.items()}
    pred = gaot_forward(params, cfg, batch["latent"], batch["xcoord"], batch["pndata"],
                        batch.get("query_coord"), batch.get("encoder_nbrs"), batch.get("decoder_nbrs"),
                        batch.get("condition"))
    loss = torch.mean((pred - batch["target"]) ** 2)
    names = [k for k in params if params[k].requires_grad]                # rotary_emb.freqs is a frozen parameter
    gs = torch.autograd.grad(loss, [params[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(params[k])) for k, g in zip(names, gs)}
    frozen = [k for k in params if not params[k].requires_grad]
    grads.update({k: torch.zeros_like(params[k]) for k in frozen})        # the reference reports no gradient for them
    if state is None:
        state = {"step": 0, "m": {k: torch.zeros_like(v) for k, v in sd.items()},
                 "v": {k: torch.zeros_like(v) for k, v in sd.items()}}
    state["step"] += 1
    new_sd = {k: sd[k] for k in frozen}                                   # torch.optim.AdamW skips parameters without a gradient
    for k in names:
        p, m, v = adamw_update(sd[k].detach(), grads[k], state["m"][k], state["v"][k], state["step"], lr, weight_decay)
        new_sd[k], state["m"][k], state["v"][k] = p, m, v
    if return_pred:
        return loss.detach(), grads, new_sd, state, pred.detach()
    return loss.detach(), grads, new_sd, state


# --------------------------------------------------------------------------------------
# autoregressive rollout (gaot.py:307-476), fx mode
# --------------------------------------------------------------------------------------
def autoregressive_predict(sd, cfg: OracleConfig, x_batch: Tensor, time_indices, t_values, stats: dict,
                           stepper_mode: str, latent: Tensor, fixed_coord: Tensor,
                           use_conditional_norm: bool = False, encoder_nbrs=None, decoder_nbrs=None) -> Tensor:
    B, N, _ = x_batch.shape
    u_mean, u_std = stats["u"]["mean"], stats["u"]["std"]
    udim = u_mean.shape[0]
    cdim = stats["c"]["mean"].shape[0] if "c" in stats else 0
    c_feat = x_batch[..., udim:udim + cdim] if cdim > 0 else None
    cur = x_batch[..., :udim]
    outs = []
    with torch.no_grad():
        for i in range(1, len(time_indices)):
            t0 = t_values[time_indices[i - 1]]
            dt = t_values[time_indices[i]] - t0
            t0n = (t0 - stats["start_time"]["mean"]) / stats["start_time"]["std"]
            dtn = (dt - stats["time_diffs"]["mean"]) / stats["time_diffs"]["std"]
            cols = [cur] + ([c_feat] if c_feat is not None else [])
            cols += [torch.full((B, N, 1), float(t0n), dtype=x_batch.dtype),
                     torch.full((B, N, 1), float(dtn), dtype=x_batch.dtype)]
            xin = torch.cat(cols, dim=-1)
            if use_conditional_norm:                                # gaot.py:403-408
                pred = gaot_forward(sd, cfg, latent, fixed_coord, xin[..., :-1], condition=xin[..., 0, -2:-1],
                                    encoder_nbrs=encoder_nbrs, decoder_nbrs=decoder_nbrs)
            else:
                pred = gaot_forward(sd, cfg, latent, fixed_coord, xin, encoder_nbrs=encoder_nbrs, decoder_nbrs=decoder_nbrs)
            if stepper_mode == "output":                            # gaot.py:454-472
                den = pred * u_std + u_mean
            elif stepper_mode == "residual":
                den = (cur * u_std + u_mean) + (pred * stats["res"]["std"] + stats["res"]["mean"])
            elif stepper_mode == "time_der":
                den = (cur * u_std + u_mean) + torch.tensor(float(dt), dtype=pred.dtype) * (
                    pred * stats["der"]["std"] + stats["der"]["mean"])
            else:
                raise ValueError(stepper_mode)
            outs.append(den)
            cur = (den - u_mean) / u_std                            # gaot.py:432
    return torch.stack(outs, dim=1)


# --------------------------------------------------------------------------------------
# weight construction with the reference's shapes (for tests that need weights but no
# golden file): same key names / shapes as GAOT.__init__ (gaot.py:21-90, magno.py:87-156,
# 423-492, attn.py:239-288).  Values are i.i.d. scaled normals, not the reference init.
# --------------------------------------------------------------------------------------
def make_state_dict(cfg: OracleConfig, input_size: int, output_size: int, seed: int = 0) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}

    def lin(name, out_f, in_f, bias=True, conv=False):
        w = torch.randn(out_f, in_f, generator=g) / math.sqrt(in_f)
        sd[f"{name}.weight"] = w[:, :, None].contiguous() if conv else w
        if bias:
            sd[f"{name}.bias"] = 0.1 * torch.randn(out_f, generator=g)

    d, C, hid = cfg.coord_dim, cfg.lifting_channels, cfg.hidden_size
    kd = d * 8 if cfg.node_embedding else d
    nonlin = cfg.transform_type in ("nonlinear", "nonlinear_kernelonly")
    for side, cin_feat, cout in (("encoder", input_size, C), ("decoder", C, C)):
        kin = 2 * kd + ((cin_feat if side == "encoder" else C) if nonlin else 0)
        sizes = [kin] + [hid] * cfg.mlp_layers + [cout]
        for i in range(len(sizes) - 1):
            lin(f"{side}.agno.channel_mlp.fcs.{i}", sizes[i + 1], sizes[i])
        if cfg.use_attention and cfg.attention_type == "dot_product":
            lin(f"{side}.agno.query_proj", 64, kd)
            lin(f"{side}.agno.key_proj", 64, kd)
        if side == "encoder":
            lin("encoder.lifting.fcs.0", C, input_size, conv=True)
        else:
            lin("decoder.projection.fcs.0", output_size, C, conv=True)
        if cfg.use_geoembed:
            if cfg.embedding_method == "statistical":
                lin(f"{side}.geoembed.mlp.0", 64, 3 + 2 * d)
                lin(f"{side}.geoembed.mlp.2", C, 64)
            else:
                lin(f"{side}.geoembed.pointnet_mlp.0", 64, d)
                lin(f"{side}.geoembed.pointnet_mlp.2", 64, 64)
                lin(f"{side}.geoembed.fc.0", C, 64)
            lin(f"{side}.recovery.fcs.0", C, 2 * C, conv=True)
        if cfg.use_scale_weights:
            lin(f"{side}.scale_weighting.0", hid // 4, kd)
            lin(f"{side}.scale_weighting.2", len(cfg.scales), hid // 4)
    # NB: with nonlinear transforms the encoder kernel MLP sees the *lifted* features? No:
    # magno.py:112-113 adds `in_channels` (the raw input width) for the encoder while AGNO is fed
    # the lifted features (magno.py:331-336) -- shapes only agree when in_channels == lifting_channels.
    tok = (cfg.patch_size ** d) * C
    D = cfg.tf_hidden_size
    lin("patch_linear", tok, tok)
    if tok != D:
        lin("processor.input_proj", D, tok)
        lin("processor.output_proj", tok, D)
    half = cfg.num_layers // 2
    blocks = [f"processor.encoder_layers.{i}" for i in range(half)]
    if cfg.num_layers % 2 == 1:
        blocks.append("processor.middle_layer")
    blocks += [f"processor.decoder_layers.{i}" for i in range(half)]
    kvd = (D // cfg.num_heads) * cfg.num_kv_heads
    for b in blocks:
        lin(f"{b}.attn.q_proj", D, D, bias=False)
        lin(f"{b}.attn.k_proj", kvd, D, bias=False)
        lin(f"{b}.attn.v_proj", kvd, D, bias=False)
        lin(f"{b}.attn.o_proj", D, D, bias=False)
        if cfg.positional_embedding == "rope":
            hd = D // cfg.num_heads
            sd[f"{b}.attn.rotary_emb.freqs"] = 1.0 / (10000.0 ** (torch.arange(0, hd, 2)[:hd // 2].float() / hd))
        lin(f"{b}.ffn.w1", D * cfg.ffn_multiplier, D, bias=False)
        lin(f"{b}.ffn.w2", D, D * cfg.ffn_multiplier, bias=False)
        lin(f"{b}.ffn.w3", D * cfg.ffn_multiplier, D, bias=False)
        if cfg.use_attn_norm:
            sd[f"{b}.attn_norm.weight"] = 1 + 0.1 * torch.randn(D, generator=g)
        if cfg.use_ffn_norm:
            sd[f"{b}.ffn_norm.weight"] = 1 + 0.1 * torch.randn(D, generator=g)
        if cfg.use_conditional_norm:
            for part in ("attn", "ffn"):
                for mlp in ("mlp_scale", "mlp_bias"):
                    sd[f"{b}.{part}.correction.{mlp}.layers.0.weight"] = 0.1 * torch.randn(D, 1, generator=g)
                    sd[f"{b}.{part}.correction.{mlp}.layers.0.bias"] = 0.1 * torch.randn(D, generator=g)
        if b.startswith("processor.decoder_layers"):
            lin(f"{b}.skip_proj", D, 2 * D)
    return sd
