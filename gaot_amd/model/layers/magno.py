"""MAGNO encoder / decoder (reference magno.py) on the HIP GNO kernels.

Differences in *how* (not what) it computes:
  * geometry-only work (int32/transposed CSR, kernel-MLP input rows, cosine attention, geometry statistics)
    lives in a GeometryPlan cached on the neighbour dict instead of being redone every forward;
  * the `cat([agno, geoembed]) -> recovery` Conv1d is evaluated as
        agno @ Wr[:, :C]^T + (geoembed @ Wr[:, C:]^T + b)        (second term is batch independent, [Q, C])
    i.e. one GEMM over B*Q rows with a Q-periodic row bias; the [B, Q, 2C] concat is never materialised;
  * tensors stay channels-last; no permutes around the point-wise convolutions.
"""
from dataclasses import dataclass, field
from typing import List, Literal, Optional, Union

import torch
import torch.nn as nn

from ... import ops
from ... import plan as _plan
from ...plan import merged_geometry
from .agno import AGNO
from .gemb import GeometricEmbedding, node_pos_encode
from .mlp import ChannelMLP
from .utils.edge_drop import apply_edge_drop_csr
from .utils.neighbor_search import NeighborSearch


@dataclass
class MAGNOConfig:
    # field names/defaults are the operator API (reference magno.py:31-60)
    coord_dim: int = 2
    radius: float = 0.033
    hidden_size: int = 64
    mlp_layers: int = 3
    lifting_channels: int = 32
    scales: List[float] = field(default_factory=lambda: [1.0])
    use_scale_weights: bool = False
    use_attention: bool = True
    attention_type: str = 'cosine'
    use_geoembed: bool = True
    embedding_method: str = 'statistical'
    pooling: str = 'max'
    transform_type: str = 'linear'
    sampling_strategy: Optional[str] = None
    max_neighbors: Optional[int] = None
    sample_ratio: Optional[float] = None
    node_embedding: bool = False
    neighbor_search_method: str = 'auto'
    use_torch_scatter: bool = True
    neighbor_strategy: str = 'radius'
    precompute_edges: bool = False

    def __post_init__(self):
        if self.coord_dim not in [2, 3]:
            raise ValueError(f"coord_dim must be 2 or 3, got {self.coord_dim}")
        if self.sampling_strategy == 'ratio' and (self.sample_ratio is None or not 0 < self.sample_ratio <= 1):
            raise ValueError("sample_ratio must be in (0, 1] when using 'ratio' sampling")
        if self.sampling_strategy == 'max_neighbors' and (self.max_neighbors is None or self.max_neighbors <= 0):
            raise ValueError("max_neighbors must be > 0 when using 'max_neighbors' sampling")


class RenumberedLists(list):
    """fx neighbour lists over the latent grid in patch-major order (model/gaot.py GAOT._patch_major, plan.renumbered): used as given,
    whether the module owns its lists or the caller handed them in"""


class _MAGNOBase(nn.Module):
    """Pieces shared by encoder and decoder: neighbour cache, per-(geometry, scale) transform, scale mixing."""

    def _setup(self, config: MAGNOConfig, feat_channels: int, kernel_out: int, kernel_extra_in: int):
        self.config = config
        self.coord_dim = config.coord_dim
        self.scales = config.scales
        self.use_scale_weights = config.use_scale_weights
        self.precompute_edges = config.precompute_edges
        self.use_geoembed = config.use_geoembed
        self.node_embedding = config.node_embedding
        self.nb_search = NeighborSearch(method=config.neighbor_search_method)
        self.neighbor_cache = {}
        self._coord_enc_cache = {}
        self._infer_cache = {}       # no_grad only: geoembed row-bias per (plan, parameter versions) -- rollouts reuse it
        self._static_unions = {}     # vx training: padded unions in static buffers per (scale, batch shape, edge bucket) -- plan.StaticUnion
        self._vx_preloaded = None    # (unions, neighbour list object): tables already uploaded for the next forward (trainer.TrainStep, autograph)
        self.sampling_strategy = config.sampling_strategy
        self.max_neighbors = config.max_neighbors
        self.sample_ratio = config.sample_ratio
        kd = self._compute_kernel_coord_dim()
        kin = 2 * kd + (kernel_extra_in if config.transform_type in ("nonlinear", "nonlinear_kernelonly") else 0)
        sizes = [kin] + [config.hidden_size] * config.mlp_layers + [kernel_out]
        self.agno = AGNO(channel_mlp_layers=sizes, transform_type=config.transform_type, use_attn=config.use_attention,
                         attention_type=config.attention_type, coord_dim=kd, use_torch_scatter=config.use_torch_scatter)
        return kd

    def _finish_setup(self, config: MAGNOConfig, kd: int, feat_channels: int):
        if self.use_geoembed:
            self.geoembed = GeometricEmbedding(input_dim=self.coord_dim, output_dim=feat_channels,
                                               method=config.embedding_method, pooling=config.pooling)
            self.recovery = ChannelMLP(in_channels=2 * feat_channels, out_channels=feat_channels, n_layers=1)
        if self.use_scale_weights:
            self.scale_weighting = nn.Sequential(nn.Linear(kd, config.hidden_size // 4), nn.ReLU(),
                                                 nn.Linear(config.hidden_size // 4, len(self.scales)))
            self.scale_weight_activation = nn.Softmax(dim=-1)

    def _compute_kernel_coord_dim(self) -> int:
        return self.coord_dim * 4 * 2 if self.node_embedding else self.coord_dim

    @staticmethod
    def _mode(coord: torch.Tensor, what: str) -> Literal['fx', 'vx']:
        if coord.ndim == 2:
            return 'fx'
        if coord.ndim == 3:
            return 'vx'
        raise ValueError(f"{what} must be 2D or 3D tensor, got shape {coord.shape}")

    def _search(self, key: str, data, queries, mode):
        """radius graphs per scale, cached by SHAPE like the reference (magno.py:177-180, 513-516)."""
        if key in self.neighbor_cache:
            return self.neighbor_cache[key]
        if mode == 'fx':
            nbrs = [self.nb_search(data=data, queries=queries, radius=self.config.radius * s) for s in self.scales]
        else:
            B = data.shape[0] if data.ndim == 3 else queries.shape[0]
            nbrs = [[self.nb_search(data=data[b] if data.ndim == 3 else data,
                                    queries=queries[b] if queries.ndim == 3 else queries,
                                    radius=self.config.radius * s) for s in self.scales] for b in range(B)]
        self.neighbor_cache[key] = nbrs
        return nbrs

    # ---- vx training on static padded unions (plan.StaticUnion): the batch composition may change every step, the launches do not
    MAX_STATIC_UNIONS = 12

    def vx_static_ok(self, feats: torch.Tensor) -> bool:
        """training passes over caller-supplied per-sample graphs (static_trainer.py:180-202) without encoded kernel coordinates; everything
        else (evaluation, rollouts, module-owned lists) keeps the composed unions of plan.merged_geometry"""
        return (_plan.VX_STATIC and feats.is_cuda and torch.is_grad_enabled() and self.training and bool(self.precompute_edges)
                and not self.node_embedding)

    def vx_unions(self, nbrs, src: torch.Tensor, dst: torch.Tensor, B: int, load: bool = True):
        """the static union of every scale for this batch, its table uploaded (not capturable).  src / dst: [B, n, d] or [n, d] coordinates.
        load=False: only find (or make) the unions; `union.load_pending(src, dst)` uploads the tables later."""
        if not isinstance(nbrs, (list, tuple)) or len(nbrs) != B:
            raise ValueError(f"vx mode: expected {B} per-sample neighbour lists")
        n_src, n_dst = int(src.shape[-2]), int(dst.shape[-2])
        src, dst = src.contiguous(), dst.contiguous()
        out = []
        for si in range(len(self.scales)):
            parts = [nbrs[b][si] for b in range(B)]
            # Dicts that come back (a dataset kept on the device) get a plan of their own on their SECOND sight and the union is composed from the
            # plans (one 9 us launch); dicts seen for the first time -- every step, when the trainer uploads the graphs per step as the reference's
            # does (move_to_device, static_trainer.py:192-193) -- are described to the compose kernel as they are (raw int64 lists): no plan is
            # built for a sample that never returns, the transposed CSR is derived on the device if a kernel asks for it.
            planned = all(_plan.has_plan(nb, n_src) for nb in parts)
            if not planned:
                seen = all(nb.get("_gaot_amd_seen") for nb in parts)
                for nb in parts:
                    nb["_gaot_amd_seen"] = True
                planned = seen
            if planned:
                items = [_plan.plan_for(nb, n_src, validate="lazy") for nb in parts]
                total = sum(p.E for p in items)
            else:
                items = parts
                total = sum(int(nb["neighbors_index"].numel()) for nb in parts)
            key = (si, B, n_src, n_dst, int(src.shape[-1]), int(dst.shape[-1]), _plan.edge_bucket(total), src.device)
            su = self._static_unions.pop(key, None)
            if su is None:
                while len(self._static_unions) >= self.MAX_STATIC_UNIONS:        # least recently used; graphs that captured it keep their reference
                    self._static_unions.pop(next(iter(self._static_unions)))
                su = _plan.StaticUnion(B, n_src, n_dst, key[4], key[5], key[6], src.device)
            self._static_unions[key] = su
            if load:
                (su.load if planned else su.load_raw)(items, src, dst)
            else:
                su.pending = items
            su.pending_raw = not planned
            out.append(su)
        return out

    def _kcoord(self, c: torch.Tensor) -> torch.Tensor:
        """kernel coordinates: raw or node_pos_encode()d (cached per tensor identity: geometry only)."""
        if not self.node_embedding:
            return c
        key = (id(c), c._version)
        hit = self._coord_enc_cache.get(key)
        if hit is None:
            if len(self._coord_enc_cache) > 64:
                self._coord_enc_cache.clear()
            hit = (c, node_pos_encode(c))
            self._coord_enc_cache[key] = hit
        if torch.cuda.is_current_stream_capturing():          # (read by address from the graph being captured: keep it past the cache entry)
            self.__dict__.setdefault("_graph_keep", []).append(hit)
        return hit[1]

    def _transform(self, src_coord, dst_coord, feats, neighbors, stats=None, head=None, lift=None, drop=True):
        """AGNO (+ geoembed + recovery) for ONE geometry at ONE scale.  feats [B, n_src, C] -> [B, n_dst, C].
        `head` = (W [out, C], b [out]) of a following point-wise linear layer (the decoder's projection): recovery and
        head are both linear with nothing in between, so they are applied as ONE map
            agno @ (W Wr1)^T + (rowb @ W^T + b)
        and the [B, n_dst, C] recovery output (33.5 MB at 16k nodes x 8) is never produced."""
        nb = neighbors
        if drop and self.training and self.sampling_strategy is not None:
            if src_coord.is_cuda:
                # neighbour sub-sampling on the device: the sub-sampled graph lives in static buffers of the full graph's capacity, re-drawn per
                # pass by kernels with a device-resident seed (plan.DropPlan) -- no host value depends on the draw, a captured step draws anew
                base = _plan.plan_for(neighbors, src_coord.shape[0])
                dp = _plan.dropped_plan(base, self.sampling_strategy, self.max_neighbors, self.sample_ratio)
                nb = neighbors if dp is base else dp.neighbors
            else:
                nb = apply_edge_drop_csr(neighbors, self.sampling_strategy, self.max_neighbors, self.sample_ratio, self.training)
        proj = rowb = w_agno = kvals = None
        if self.use_geoembed:
            w = self.recovery.fcs[0].weight.squeeze(-1)                                             # [C, 2C]
            C = w.shape[0]
            # one gradient assembly instead of two slice-backward chains; with a registered gradient slice the two halves are
            # written in place by the layers that use them
            w_agno, w_geo = ops.split_cols(w, C, param=self.recovery.fcs[0].weight)
            key = None
            if not torch.is_grad_enabled() and nb is neighbors:
                # inference (autoregressive rollouts): geometry and weights are fixed across steps -> keep the row bias
                key = (id(nb), id(src_coord), src_coord._version, id(dst_coord), dst_coord._version,
                       tuple(p._version for p in self.geoembed.parameters()), self.recovery.fcs[0].weight._version,
                       self.recovery.fcs[0].bias._version, ops.weights_generation())
                hit = self._infer_cache.get("rowb")
                if hit is not None and hit[0] == key and hit[1] is nb:
                    rowb = hit[2]
                    if torch.cuda.is_current_stream_capturing():
                        # the graph being captured reads this cached tensor by address: it must outlive the cache entry (another geometry's
                        # evaluation pass replaces the entry; the replay would read freed memory)
                        self.__dict__.setdefault("_graph_keep", []).append(hit)
            if rowb is None:
                # embedding MLP and the geoembed half of the recovery block as one chain: [n_dst, C] -- in training together with the kernel
                # MLP of this transform, ONE launch each way (ops.mlp_chain_pair: the chain alone is a few dozen workgroups on 256 CUs)
                geo = kern = None
                if torch.is_grad_enabled() and src_coord.is_cuda:
                    geo = self.geoembed.chain_args(src_coord, dst_coord, nb, stats if nb is neighbors else None, (w_geo, self.recovery.fcs[0].bias))
                    kern = self.agno.kernel_chain_args(self._kcoord(src_coord), nb, self._kcoord(dst_coord)) if geo is not None else None
                if geo is not None and kern is not None:
                    kvals, rowb = ops.mlp_chain_pair(*kern, *geo)
                else:
                    rowb = self.geoembed(input_geom=src_coord, latent_queries=dst_coord, spatial_nbrs=nb,
                                         stats=stats if nb is neighbors else None, head=(w_geo, self.recovery.fcs[0].bias))
                if key is not None:
                    self._infer_cache["rowb"] = (key, nb, rowb, (src_coord, dst_coord))      # hold the tensors: ids stay unique
            if head is not None:
                hw, hb = head
                weff, rproj = ops.proj_fold(hw, hb, w_agno, rowb)      # [out, C] = W @ Wr1 and [n_dst, out] = rowb @ W^T + b (tiny)
                proj = (weff, rproj, None)
        elif head is not None:
            proj = (head[0], None, head[1])
        # with few output channels the folded map is applied INSIDE the transform kernels (AGNO decides: `applied_proj`)
        out = self.agno(y=self._kcoord(src_coord), x=self._kcoord(dst_coord), f_y=feats, neighbors=nb, lift=lift, proj=proj, kernel_values=kvals)
        if proj is not None:
            return out if self.agno.applied_proj else ops.linear(out, proj[0], proj[2], rowbias=proj[1])
        if self.use_geoembed:
            out = ops.linear(out, w_agno, rowbias=rowb, publish=True)      # (the encoder's tokens: the patch embedding reads them through a reshape)
        return out

    def _scale_mix_weights(self, coords: torch.Tensor) -> torch.Tensor:
        h = ops.mlp_chain(coords, [self.scale_weighting[0].weight, self.scale_weighting[2].weight],
                          [self.scale_weighting[0].bias, self.scale_weighting[2].bias], ["relu", "none"])
        return self.scale_weight_activation(h)

    def _combine(self, per_scale: List[torch.Tensor], weights: Optional[torch.Tensor]) -> torch.Tensor:
        if len(per_scale) == 1:
            return per_scale[0]
        return ops.scale_mix(per_scale, weights if self.use_scale_weights else None)

    def _all_scales(self, mode, src, dst, feats, nbrs, head=None, lift=None):
        """`lift` = (pn, W, b) replaces `feats` = W pn + b (encoder): the lifting is folded into the transform kernels."""
        per_scale = []
        for si in range(len(self.scales)):
            if mode == 'fx':
                per_scale.append(self._transform(src, dst, feats, nbrs[si], head=head, lift=lift))
            else:
                # vx: block-diagonal union of the B per-sample graphs -> one launch per kernel for the whole batch
                if lift is not None:
                    feats = lift[0]
                B = feats.shape[0]
                if self.vx_static_ok(feats):
                    if si == 0:
                        pre, self._vx_preloaded = self._vx_preloaded, None
                        unions = pre[0] if (pre is not None and pre[1] is nbrs) else self.vx_unions(nbrs, src, dst, B)
                    mg = unions[si]
                    if mg.n_src != feats.shape[1]:
                        raise ValueError("vx mode needs the same number of source / query points in every sample of a batch")
                    mg.refresh()
                    pl, nbu = mg.plan, mg.neighbors
                    if self.sampling_strategy is not None:
                        # drawn per edge / per row, so drawing on the union IS drawing per sample (magno.py:372-378); the statistics of the
                        # sub-sampled graph are standardised per sample (gemb.py:164-169) like the full graph's
                        pl = _plan.dropped_plan(mg.plan, self.sampling_strategy, self.max_neighbors, self.sample_ratio)
                        nbu = mg.neighbors if pl is mg.plan else pl.neighbors
                    stats = pl.geo_stats(mg.src, mg.dst, groups=B) if (self.use_geoembed and self.geoembed.method == 'statistical') else None
                    flat = feats.reshape(1, B * feats.shape[1], feats.shape[2])
                    if lift is not None:
                        out = self._transform(mg.src, mg.dst, None, nbu, stats, head=head, lift=(flat, lift[1], lift[2]), drop=False)
                    else:
                        out = self._transform(mg.src, mg.dst, flat, nbu, stats, head=head, drop=False)
                    per_scale.append(out.reshape(B, mg.n_dst, out.shape[-1]))
                    continue
                srcs = [src[b] if src.ndim == 3 else src for b in range(B)]
                dsts = [dst[b] if dst.ndim == 3 else dst for b in range(B)]
                parts = [nbrs[b][si] for b in range(B)]
                if self.training and self.sampling_strategy is not None:
                    # neighbour sub-sampling is drawn PER SAMPLE (magno.py:372-378), and the geometry statistics of the dropped
                    # graph are standardised per sample (gemb.py:164-169): drop first, then merge the dropped graphs
                    parts = [apply_edge_drop_csr(nb_, self.sampling_strategy, self.max_neighbors, self.sample_ratio, True) for nb_ in parts]
                mg = merged_geometry(parts, srcs, dsts, parents=(src, dst))
                n_dst = mg.n_dst_each[0]
                if any(n != n_dst for n in mg.n_dst_each) or any(n != feats.shape[1] for n in mg.n_src_each):
                    raise ValueError("vx mode needs the same number of source / query points in every sample of a batch")
                stats = mg.geo_stats() if (self.use_geoembed and self.geoembed.method == 'statistical') else None
                flat = feats.reshape(1, B * feats.shape[1], feats.shape[2])
                if lift is not None:
                    out = self._transform(mg.src, mg.dst, None, mg.neighbors, stats, head=head, lift=(flat, lift[1], lift[2]), drop=False)
                else:
                    out = self._transform(mg.src, mg.dst, flat, mg.neighbors, stats, head=head, drop=False)
                per_scale.append(out.reshape(B, n_dst, out.shape[-1]))
        return per_scale


class MAGNOEncoder(_MAGNOBase):
    """physical nodes -> latent tokens."""

    def __init__(self, in_channels: int, out_channels: int, config: MAGNOConfig):
        super().__init__()
        kd = self._setup(config, out_channels, kernel_out=out_channels, kernel_extra_in=in_channels)
        self.lifting = ChannelMLP(in_channels=in_channels, hidden_channels=config.hidden_size,
                                  out_channels=out_channels, n_layers=1)
        self._finish_setup(config, kd, out_channels)

    def _detect_coordinate_mode(self, x_coord):
        return self._mode(x_coord, "x_coord")

    def _compute_neighbors(self, x_coord, latent_coord, mode):
        key = f"{mode}_{x_coord.shape}_{latent_coord.shape}_{tuple(self.scales)}"
        return self._search(key, x_coord, latent_coord, mode)

    def forward(self, x_coord: torch.Tensor, pndata: torch.Tensor, latent_tokens_coord: torch.Tensor,
                encoder_nbrs: Optional[Union[List, List[List]]] = None) -> torch.Tensor:
        mode = self._detect_coordinate_mode(x_coord)
        B = pndata.shape[0]
        if mode == 'fx':
            if x_coord.shape[1] != self.coord_dim:
                raise ValueError(f"Expected x_coord shape [num_nodes, {self.coord_dim}], got {x_coord.shape}")
            n = x_coord.shape[0]
        else:
            if x_coord.shape[0] != B or x_coord.shape[2] != self.coord_dim:
                raise ValueError(f"Expected x_coord shape [{B}, num_nodes, {self.coord_dim}], got {x_coord.shape}")
            n = x_coord.shape[1]
        if tuple(pndata.shape[:2]) != (B, n):
            raise ValueError(f"pndata shape mismatch: expected [{B}, {n}, in_channels], got {pndata.shape}")
        if latent_tokens_coord.shape[1] != self.coord_dim:
            raise ValueError(f"Expected latent_tokens_coord shape [num_latent, {self.coord_dim}], got {latent_tokens_coord.shape}")
        if self.precompute_edges:
            if encoder_nbrs is None:
                raise ValueError("encoder_nbrs required when precompute_edges=True")
            nbrs = encoder_nbrs
        elif isinstance(encoder_nbrs, RenumberedLists):
            nbrs = encoder_nbrs
        else:
            nbrs = self._compute_neighbors(x_coord, latent_tokens_coord, mode)
        w = self._scale_mix_weights(self._kcoord(latent_tokens_coord)) if self.use_scale_weights else None
        if self.lifting.n_layers == 1:
            # one point-wise linear map: folded into the integral transform (AGNO decides; falls back to forming it)
            lift = (pndata, self.lifting.fcs[0].weight, self.lifting.fcs[0].bias)
            return self._combine(self._all_scales(mode, x_coord, latent_tokens_coord, None, nbrs, lift=lift), w)
        lifted = self.lifting.forward_channels_last(pndata)                                     # [B, n, C]
        return self._combine(self._all_scales(mode, x_coord, latent_tokens_coord, lifted, nbrs), w)


class MAGNODecoder(_MAGNOBase):
    """latent tokens -> query nodes, then the output projection."""

    def __init__(self, in_channels: int, out_channels: int, config: MAGNOConfig):
        super().__init__()
        kd = self._setup(config, in_channels, kernel_out=in_channels, kernel_extra_in=in_channels)
        self.projection = ChannelMLP(in_channels=in_channels, hidden_channels=config.hidden_size,
                                     out_channels=out_channels, n_layers=1)
        self._finish_setup(config, kd, in_channels)

    def _detect_coordinate_mode(self, query_coord):
        return self._mode(query_coord, "query_coord")

    def _compute_neighbors(self, latent_coord, query_coord, mode):
        key = f"dec_{mode}_{latent_coord.shape}_{query_coord.shape}_{tuple(self.scales)}"
        return self._search(key, latent_coord, query_coord, mode)

    def forward(self, latent_tokens_coord: torch.Tensor, rndata: torch.Tensor, query_coord: torch.Tensor,
                decoder_nbrs: Optional[Union[List, List[List]]] = None) -> torch.Tensor:
        mode = self._detect_coordinate_mode(query_coord)
        B = rndata.shape[0]
        if mode == 'fx':
            if query_coord.shape[1] != self.coord_dim:
                raise ValueError(f"Expected query_coord shape [num_query, {self.coord_dim}], got {query_coord.shape}")
        elif query_coord.shape[0] != B or query_coord.shape[2] != self.coord_dim:
            raise ValueError(f"Expected query_coord shape [{B}, num_query, {self.coord_dim}], got {query_coord.shape}")
        if latent_tokens_coord.shape[1] != self.coord_dim:
            raise ValueError(f"Expected latent_tokens_coord shape [num_latent, {self.coord_dim}], got {latent_tokens_coord.shape}")
        if self.precompute_edges:
            if decoder_nbrs is None:
                raise ValueError("decoder_nbrs required when precompute_edges=True")
            nbrs = decoder_nbrs
        elif isinstance(decoder_nbrs, RenumberedLists):
            nbrs = decoder_nbrs
        else:
            nbrs = self._compute_neighbors(latent_tokens_coord, query_coord, mode)
        w = None
        if self.use_scale_weights:        # vx: the FIRST sample's coordinates (reference magno.py:610-612)
            w = self._scale_mix_weights(self._kcoord(query_coord if mode == 'fx' else query_coord[0]))
        if len(self.scales) == 1 and self.projection.n_layers == 1:
            # single scale: fold the output projection into the recovery map (see _transform)
            pw = self.projection.fcs[0].weight.squeeze(-1)
            return self._all_scales(mode, latent_tokens_coord, query_coord, rndata, nbrs, head=(pw, self.projection.fcs[0].bias))[0]
        dec = self._combine(self._all_scales(mode, latent_tokens_coord, query_coord, rndata, nbrs), w)
        return self.projection.forward_channels_last(dec)
