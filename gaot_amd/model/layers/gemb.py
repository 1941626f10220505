"""Geometric embedding (reference gemb.py) on the HIP geometry-statistics kernel + GEMM MLP."""
import math

import torch
import torch.nn as nn

from ... import ops
from ...plan import plan_for


def node_pos_encode(x: torch.Tensor, freq: int = 4) -> torch.Tensor:
    """sin/cos features of pi*(x+1) at integer frequencies 1..freq -> [n, freq*2*d] (gemb.py:12-34).
    Geometry-only (no parameters, no grad); evaluated once per geometry with torch elementwise ops."""
    assert x.ndim == 2, f"The x is expected to be 2D tensor, but got shape {x.shape}"
    k = torch.arange(1, freq + 1, device=x.device)
    ang = k[None, :, None] * (math.pi * (x + 1))[:, None, :]
    return torch.cat([ang.sin(), ang.cos()], dim=2).reshape(x.shape[0], -1)


class GeometricEmbedding(nn.Module):
    def __init__(self, input_dim, output_dim, method='statistical', pooling='max', **kwargs):
        super().__init__()
        self.input_dim = input_dim
        self.output_dim = output_dim
        self.method = method.lower()
        self.pooling = pooling.lower()
        if self.pooling not in ('max', 'mean'):
            raise ValueError(f"Unsupported pooling method: {self.pooling}. Supported methods: 'max', 'mean'.")
        if self.method == 'statistical':
            self.mlp = nn.Sequential(nn.Linear(3 + 2 * input_dim, 64), nn.ReLU(), nn.Linear(64, output_dim), nn.ReLU())
        elif self.method == 'pointnet':
            self.pointnet_mlp = nn.Sequential(nn.Linear(input_dim, 64), nn.ReLU(), nn.Linear(64, 64), nn.ReLU())
            self.fc = nn.Sequential(nn.Linear(64, output_dim), nn.ReLU())
        else:
            raise ValueError(f"Unknown method: {self.method}")

    def chain_args(self, input_geom, latent_queries, spatial_nbrs, stats=None, head=None):
        """(x, weights, biases, acts) of the statistical branch with its `head` as ONE row-wise chain (what forward() hands to
        ops.mlp_chain), or None for the other branches: the caller may pair it with another chain in one launch (ops.mlp_chain_pair)"""
        if self.method != 'statistical' or head is None:
            return None
        if stats is None:
            stats = plan_for(spatial_nbrs, input_geom.shape[0]).geo_stats(input_geom, latent_queries)
        return (stats, [self.mlp[0].weight, self.mlp[2].weight, head[0]], [self.mlp[0].bias, self.mlp[2].bias, head[1]], ["relu", "relu", "none"])

    def forward(self, input_geom, latent_queries, spatial_nbrs, stats=None, head=None):
        """`head` = (W [C_out, output_dim], b [C_out]): a Linear applied to the embedding by the caller (the geoembed half of
        the recovery block, magno.py:345-350); with it the statistical branch is ONE chain relu, relu, none that the fused
        row-wise MLP kernels serve when every width is 64."""
        plan = plan_for(spatial_nbrs, input_geom.shape[0])
        if self.method == 'statistical':
            if stats is None:
                stats = plan.geo_stats(input_geom, latent_queries)      # [Q, 3+2d], cached per geometry
            if head is not None:
                return ops.mlp_chain(stats, [self.mlp[0].weight, self.mlp[2].weight, head[0]],
                                     [self.mlp[0].bias, self.mlp[2].bias, head[1]], ["relu", "relu", "none"])
            return ops.mlp_chain(stats, [self.mlp[0].weight, self.mlp[2].weight],
                                 [self.mlp[0].bias, self.mlp[2].bias], ["relu", "relu"])
        emb = self._pointnet(plan, input_geom, latent_queries)
        return emb if head is None else ops.linear(emb, head[0], head[1])

    def _pointnet(self, plan, geom, queries):
        """Per-edge MLP on centred neighbour coordinates, pooled per query (gemb.py:173-228)."""
        Q = queries.shape[0]
        out = torch.zeros(Q, self.output_dim, device=geom.device, dtype=torch.float32)
        if plan.E == 0:
            return out
        feat = plan.edge_features(geom, queries)
        d = geom.shape[1]
        rel = feat[:, :d] - feat[:, d:]
        h = ops.mlp_chain(rel, [self.pointnet_mlp[0].weight, self.pointnet_mlp[2].weight],
                          [self.pointnet_mlp[0].bias, self.pointnet_mlp[2].bias], ["relu", "relu"])
        if self.pooling == 'mean':
            pooled = ops.segment_sum(h[None], plan, 1.0 / plan.deg.clamp(min=1).to(torch.float32))[0]
        else:
            pooled = ops.segment_max(h, plan)
        emb = ops.mlp_chain(pooled, [self.fc[0].weight], [self.fc[0].bias], ["relu"])
        return torch.where((plan.deg > 0)[:, None], emb, out)
