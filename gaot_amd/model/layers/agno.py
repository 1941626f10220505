"""Attentional graph neural operator (reference agno.py) on the HIP gather / segment-reduce kernels.

    out[b,i,:] = sum_{e in N(i)} a_e * k_e (*) f[b, j(e), :]
with k_e = MLP([y_j, x_i]) (GEMM chain over the E edge rows, batch independent for the default 'linear'
transform) and a_e the per-edge scalar: cosine segment-softmax (geometry only -> cached in the plan),
learned dot-product segment-softmax, quadrature weight, or 1/deg for the plain mean.
"""
import math
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops
from ...plan import plan_for
from .mlp import LinearChannelMLP

_TRANSFORMS = ("linear_kernelonly", "linear", "nonlinear_kernelonly", "nonlinear")


class AGNO(nn.Module):
    def __init__(self, channel_mlp=None, channel_mlp_layers=None, channel_mlp_non_linearity=F.gelu,
                 transform_type="linear", use_attn=None, attention_type='cosine', coord_dim=None, use_torch_scatter=True):
        super().__init__()
        if channel_mlp is None and channel_mlp_layers is None:
            raise ValueError("Either channel_mlp or channel_mlp_layers must be provided.")
        if transform_type not in _TRANSFORMS:
            raise ValueError(f"Invalid transform_type: {transform_type}")
        self.transform_type = transform_type
        self.use_attn = use_attn
        self.attention_type = attention_type
        self.use_torch_scatter = use_torch_scatter     # accepted for signature compatibility; unused
        if use_attn:
            if coord_dim is None:
                raise ValueError("coord_dim must be specified when use_attn is True")
            if attention_type not in ('cosine', 'dot_product'):
                raise ValueError(f"Invalid attention_type: {attention_type}")
            self.coord_dim = coord_dim
        self.channel_mlp = channel_mlp if channel_mlp is not None else LinearChannelMLP(layers=channel_mlp_layers, non_linearity=channel_mlp_non_linearity)
        if use_attn and attention_type == 'dot_product':
            self.query_proj = nn.Linear(coord_dim, 64)
            self.key_proj = nn.Linear(coord_dim, 64)
            self.scaling_factor = 1.0 / math.sqrt(64.0)

    # per-edge scalar a_e (None => plain sum)
    def _edge_scale(self, plan, y, x, weights):
        a = None
        if self.use_attn:
            # the reference slices [:coord_dim] (agno.py:212-213); coord_dim is the full kernel-coordinate width,
            # so the slice is normally the tensor itself (kept identical for the plan's identity-keyed cache)
            ys = y if y.shape[1] == self.coord_dim else y[:, :self.coord_dim]
            xs = x if x.shape[1] == self.coord_dim else x[:, :self.coord_dim]
            if self.attention_type == 'cosine':
                a = plan.cosine_attention(ys, xs)
            else:
                # <Wq x_i + bq, Wk y_j + bk> / 8: project the NODES (small GEMMs), gather per edge, segment softmax
                qn = ops.linear(xs, self.query_proj.weight, self.query_proj.bias)
                kn = ops.linear(ys, self.key_proj.weight, self.key_proj.bias)
                a = ops.segment_softmax(ops.edge_dot_score(qn, kn, plan, self.scaling_factor), plan)
        if weights is not None:
            assert weights.ndim == 1, "Weights must be of dimension 1 in all cases"
            wq = weights[plan.index_long]
            a = wq if a is None else a * wq
        elif not self.use_attn:
            a = plan.inv_deg_edge                      # 'mean' reduction (agno.py:264)
        return a

    def kernel_chain_args(self, y: torch.Tensor, neighbors: Dict[str, torch.Tensor], x: Optional[torch.Tensor] = None):
        """(edge rows, weights, biases, acts) of the kernel MLP as forward() would run it for the batch-independent kernels (`linear`
        transforms), or None: the caller may compute k_e together with another chain (ops.mlp_chain_pair) and hand it back as
        `kernel_values`"""
        from .mlp import LinearChannelMLP
        mlp = self.channel_mlp
        if (self.transform_type not in ("linear", "linear_kernelonly") or type(mlp) is not LinearChannelMLP or mlp.act is None
                or mlp.dropout is not None):
            return None
        if x is None:
            x = y
        plan = plan_for(neighbors, y.shape[0])
        if plan.E == 0:
            return None
        acts = [mlp.act] * (mlp.n_layers - 1) + ["none"]
        return (plan.edge_features(y, x), [fc.weight for fc in mlp.fcs], [fc.bias for fc in mlp.fcs], acts)

    def forward(self, y: torch.Tensor, neighbors: Dict[str, torch.Tensor], x: Optional[torch.Tensor] = None,
                f_y: Optional[torch.Tensor] = None, weights: Optional[torch.Tensor] = None, lift=None, proj=None, kernel_values=None):
        """`lift` = (pn [B,n,c_in], W [C,c_in(,1)], b [C] or None): f_y is the point-wise LINEAR map W pn + b of raw node data
        (the encoder's lifting, magno.py:334).  When the fused kernels apply, f_y is never formed; otherwise it is computed here.
        `kernel_values` = k_e [E, C] already computed by the caller from kernel_chain_args() (`linear` transforms only)."""
        if x is None:
            x = y
        self.applied_proj = False           # set when `proj` = (weff [OC,C], rowbias [Q,OC] | None, bias [OC] | None) was folded in
        if lift is not None and f_y is None:
            pn, lw, lb = lift
            fusable = (self.transform_type == "linear" and ops._GNOLiftTransform.eligible(pn, lw, lw.shape[0], None)
                       and not (self.use_attn and self.attention_type == 'dot_product') and weights is None)
            if not fusable:
                f_y = ops.linear(pn, lw, lb)
                lift = None
        plan = plan_for(neighbors, y.shape[0])
        a = self._edge_scale(plan, y, x, weights)
        feat = plan.edge_features(y, x)                                      # [E, 2*kd]  (y_j first, then x_i)
        batched = f_y is not None and f_y.ndim == 3
        if f_y is not None and f_y.ndim not in (2, 3):
            raise ValueError(f"f_y has unexpected ndim: {f_y.ndim}")
        f3 = None if f_y is None else (f_y if batched else f_y[None])

        if f3 is not None and self.transform_type in ("nonlinear", "nonlinear_kernelonly"):
            # kernel sees f(y_j): k is [B,E,C]; a batched GEMM chain over B*E rows, then a plain segment sum
            k = self.channel_mlp(ops.edge_cat(feat, f3, plan))                # [B, E, C]: rows [y_j, x_i, f(y_j)] per sample
            out = ops.nonlinear_transform(k, f3, plan, a, self.transform_type == "nonlinear")
        else:
            k = kernel_values
            if k is None and not torch.is_grad_enabled():      # rollouts: k_e depends on geometry + weights only -> reuse across steps
                key = (id(feat), plan.epoch, tuple(p._version for p in self.channel_mlp.parameters()), ops.weights_generation())
                hit = getattr(self, "_infer_k", None)
                if hit is not None and hit[0] == key and hit[1] is feat:
                    k = hit[2]
                    if torch.cuda.is_current_stream_capturing():      # read by address from the graph being captured: keep it past the cache entry
                        self.__dict__.setdefault("_graph_keep", []).append(hit)
            if k is None:
                k = self.channel_mlp(feat)                                   # [E, C]
                if not torch.is_grad_enabled():
                    self._infer_k = (key, feat, k)
            if lift is not None:
                return ops.gno_lift_transform(k, lift[0], lift[1], lift[2], plan, a)         # [B, Q, C]
            if (proj is not None and f3 is not None and batched and self.transform_type == "linear" and weights is None
                    and not (self.use_attn and self.attention_type == 'dot_product')
                    and ops._GNOProjTransform.eligible(f3, proj[0], None)):
                self.applied_proj = True
                return ops.gno_proj_transform(k, f3, proj[0], proj[1], proj[2], plan, a)     # [B, Q, OC]
            if f3 is None:                                                   # transform (a): integrate the kernel itself
                kk = k if a is None else k * a[:plan.E, None]
                out = ops.segment_sum(kk[None], plan)
            else:
                out = ops.gno_transform(k, f3, plan, a)
        return out if batched else out[0]
