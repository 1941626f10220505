"""Parameter containers + forward through the HIP GEMM for the reference's MLP family (src/model/layers/mlp.py).

The nn.Linear / nn.Conv1d sub-modules exist so that parameter names, shapes and default initialisation (and the
RNG stream consumed at construction) equal the reference's; their own forward() is never used.
"""
from typing import Callable, Optional, Sequence

import torch
import torch.nn as nn

from ... import ops


def activation_name(fn: Optional[Callable]) -> Optional[str]:
    """the reference passes activation CALLABLES (mlp.py:253,311, default F.gelu); the fused HIP kernels know exact-erf GELU and ReLU
    by name.  Any other callable returns None: the layers then run one HIP GEMM each with the caller's own callable in between."""
    import torch.nn.functional as F
    if fn is None or fn is F.gelu or (isinstance(fn, nn.GELU) and getattr(fn, "approximate", "none") == "none"):
        return "gelu"
    if fn is F.relu or fn is torch.relu or isinstance(fn, nn.ReLU):
        return "relu"
    return None


def _generic_chain(x, fcs, non_linearity, drops):
    """mlp.py:283-298,329-337 for an arbitrary activation callable and / or dropout: every Linear / Conv1d(k=1) is a HIP GEMM
    (ops.linear, channels last), the caller's callable and nn.Dropout run between them as what they are"""
    import torch.nn.functional as F
    act = F.gelu if non_linearity is None else non_linearity
    n = len(fcs)
    for i, fc in enumerate(fcs):
        x = ops.linear(x, fc.weight, fc.bias)
        if i < n - 1:
            x = act(x)
        if drops is not None:
            x = drops[i](x)
    return x


class LinearChannelMLP(nn.Module):
    """Kernel MLP of the integral transform (reference mlp.py:307-337): Linear + activation (default GELU(erf)), last layer bare."""

    def __init__(self, layers: Sequence[int], non_linearity: Optional[Callable] = None, dropout: float = 0.0):
        super().__init__()
        self.non_linearity = non_linearity
        self.act = activation_name(non_linearity)
        if len(layers) < 2:
            raise AssertionError("LinearChannelMLP needs at least one layer")
        self.n_layers = len(layers) - 1
        self.fcs = nn.ModuleList()
        # same registration order as the reference (fcs, then dropout: mlp.py:318-327) -- dropout holds no parameters
        self.dropout = nn.ModuleList([nn.Dropout(dropout) for _ in range(self.n_layers)]) if dropout and dropout > 0.0 else None
        for i in range(self.n_layers):
            self.fcs.append(nn.Linear(layers[i], layers[i + 1]))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.act is None or self.dropout is not None:
            return _generic_chain(x, self.fcs, self.non_linearity, self.dropout)
        acts = [self.act] * (self.n_layers - 1) + ["none"]
        return ops.mlp_chain(x, [fc.weight for fc in self.fcs], [fc.bias for fc in self.fcs], acts)


class ChannelMLP(nn.Module):
    """Point-wise channel mixing (reference mlp.py:227-305).  GAOT only instantiates n_layers=1, i.e. a single
    Conv1d(k=1) without activation.  Unlike the reference, forward takes CHANNELS-LAST input [..., n, c_in]
    (`forward_channels_last`); `forward` keeps the reference's [B, c, n] convention for drop-in callers."""

    def __init__(self, in_channels, out_channels=None, hidden_channels=None, n_layers=2, n_dim=2,
                 non_linearity=None, dropout=0.0, **kwargs):
        super().__init__()
        self.n_layers = n_layers
        self.in_channels = in_channels
        self.out_channels = in_channels if out_channels is None else out_channels
        self.hidden_channels = in_channels if hidden_channels is None else hidden_channels
        self.non_linearity = non_linearity
        self.act = activation_name(non_linearity)
        self.dropout = nn.ModuleList([nn.Dropout(dropout) for _ in range(n_layers)]) if dropout and dropout > 0.0 else None
        widths = [self.in_channels] + [self.hidden_channels] * (n_layers - 1) + [self.out_channels]
        self.fcs = nn.ModuleList(nn.Conv1d(widths[i], widths[i + 1], 1) for i in range(n_layers))

    def forward_channels_last(self, x: torch.Tensor, rowbias: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.dropout is not None or (self.n_layers > 1 and self.act is None):
            y = _generic_chain(x, self.fcs, self.non_linearity, self.dropout)
            return y if rowbias is None else y + rowbias
        if self.n_layers == 1:
            return ops.linear(x, self.fcs[0].weight, self.fcs[0].bias, rowbias=rowbias)
        acts = [self.act] * (self.n_layers - 1) + ["none"]
        return ops.mlp_chain(x, [fc.weight for fc in self.fcs], [fc.bias for fc in self.fcs], acts)

    def forward(self, x: torch.Tensor) -> torch.Tensor:          # [B, c_in, n] -> [B, c_out, n]
        lead = x.shape[:2]
        y = self.forward_channels_last(x.reshape(*lead, -1).transpose(1, 2))
        return y.transpose(1, 2).reshape(lead[0], self.out_channels, *x.shape[2:])


class _SingleLinear(nn.Module):
    """`layers.0` holder so ConditionedNorm keys read `mlp_scale.layers.0.weight` (reference mlp.py:41-72)."""

    def __init__(self, fin: int, fout: int):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(fin, fout)])
        self.layers[0].reset_parameters()    # the reference's MLP re-draws after construction (mlp.py:61-65): same RNG stream


class ConditionedNorm(nn.Module):
    """x * (1 + c*Ws(c)) + c*Wb(c)  (reference mlp.py:74-124; num_layers=2 there means ONE Linear each)."""

    def __init__(self, input_size: int, output_size: int, hidden_size: int):
        super().__init__()
        self.mlp_scale = _SingleLinear(input_size, output_size)
        self.mlp_bias = _SingleLinear(input_size, output_size)
        for m in (self.mlp_scale, self.mlp_bias):      # reference order: MLP.reset_parameters, then N(0, 0.01) weights
            nn.init.normal_(m.layers[0].weight, std=0.01)

    def forward(self, c: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        # [B,1] conditioning -> [B,D] scale / shift (a few hundred flops), then one modulation pass over x
        s = 1 + c * ops.linear(c, self.mlp_scale.layers[0].weight, self.mlp_scale.layers[0].bias)
        b = c * ops.linear(c, self.mlp_bias.layers[0].weight, self.mlp_bias.layers[0].bias)
        if x.dim() == 3 and s.dim() == 2 and s.shape[0] == x.shape[0]:
            return ops.cond_affine(x, s, b)
        return x * s[:, None, :] + b[:, None, :]
