"""ViT processor of GAOT on the HIP kernels (mirrors the module/parameter layout of reference attn.py)."""
from dataclasses import dataclass, field, fields, is_dataclass
from typing import Optional

import torch
import torch.nn as nn

from ... import ops
from .mlp import ConditionedNorm


@dataclass
class AttentionConfig:
    num_heads: int = 8
    num_kv_heads: int = 8
    use_conditional_norm: bool = False
    cond_norm_hidden_size: int = 4
    atten_dropout: float = 0.0


@dataclass
class TransformerConfig:
    patch_size: int = 8
    hidden_size: int = 256
    use_attn_norm: bool = True
    use_ffn_norm: bool = True
    norm_eps: float = 1e-6
    num_layers: int = 3
    positional_embedding: str = 'absolute'
    use_long_range_skip: bool = True
    ffn_multiplier: int = 4
    attn_config: AttentionConfig = field(default_factory=AttentionConfig)


def _cfg_get(cfg, name):
    return cfg[name] if isinstance(cfg, dict) else getattr(cfg, name)


def shallow_asdict(obj) -> dict:
    if is_dataclass(obj):
        return {f.name: getattr(obj, f.name) for f in fields(obj)}
    if isinstance(obj, dict):
        return dict(obj)
    raise TypeError(f"Unsupported type for shallow_asdict: {type(obj)}")


class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return ops.rms_norm(x, self.weight, self.eps)


class RotaryEmbedding(nn.Module):
    """State and angles of `rotary_embedding_torch.RotaryEmbedding(dim=head_dim)` as the reference constructs it
    (attn.py:75-76; library defaults freqs_for='lang', theta=10000, learned_freq=False): `freqs` is a non-trainable
    nn.Parameter (so it appears in the state_dict as `...attn.rotary_emb.freqs`), position = sequence index.
    The dependency is not vendored or pinned by the reference: restated from its published source, parity unpinned."""

    def __init__(self, dim: int, theta: float = 10000.0):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[:dim // 2].float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)
        self._table = None

    def cos_sin(self, S: int, device) -> torch.Tensor:
        """[S, dim/2, 2] (cos, sin) of pos * freqs: no learnable input, built once per (S, device, freqs version)"""
        key = (S, device, self.freqs._version)
        if self._table is None or self._table[0] != key:
            ang = torch.arange(S, device=device, dtype=torch.float32)[:, None] * self.freqs.to(device)[None, :]
            self._table = (key, torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous())
        return self._table[1]


class GroupQueryFlashAttention(nn.Module):
    """q/k/v projection as ONE GEMM over the concatenated weights, fp32 flash attention, o_proj with the block's
    residual fused into its epilogue (attn.py:78-119)."""

    def __init__(self, input_size: int, hidden_size: int, num_heads: int = 8, num_kv_heads: int = 8,
                 use_conditional_norm: bool = False, cond_norm_hidden_size: int = 4, atten_dropout: float = 0.0,
                 positional_embedding: str = "absolute"):
        super().__init__()
        assert hidden_size % num_heads == 0, f"hidden_size {hidden_size} must be divisible by num_heads {num_heads}"
        assert num_heads % num_kv_heads == 0, f"num_heads {num_heads} must be divisible by num_kv_heads {num_kv_heads}"
        self.num_heads = num_heads
        self.num_kv_heads = num_kv_heads
        self.num_repeat = num_heads // num_kv_heads
        self.head_dim = hidden_size // num_heads
        self.atten_dropout = atten_dropout
        kv = self.head_dim * num_kv_heads
        self.q_proj = nn.Linear(input_size, hidden_size, bias=False)
        self.k_proj = nn.Linear(input_size, kv, bias=False)
        self.v_proj = nn.Linear(input_size, kv, bias=False)
        self.o_proj = nn.Linear(hidden_size, input_size, bias=False)
        self.correction = ConditionedNorm(1, input_size, cond_norm_hidden_size) if use_conditional_norm else None
        if positional_embedding == "rope":
            if self.head_dim % 2:
                raise ValueError("rope needs an even head_dim")
            self.rotary_emb = RotaryEmbedding(dim=self.head_dim)
        if self.head_dim > 128:
            raise NotImplementedError("the attention kernels support head_dim <= 128")

    def forward(self, x, condition=None, relative_positions=None, residual=None):
        if self.correction is not None:
            x = self.correction(c=condition, x=x)
        if x.is_cuda:
            ops.adopt_adjacent_storage([self.q_proj.weight, self.k_proj.weight, self.v_proj.weight])
        qkv = ops.linear_cat(x, [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight])
        if relative_positions is not None:       # the reference only tests for None: the rotation angle is the SEQUENCE index
            qkv = ops.rope(qkv, self.num_heads + self.num_kv_heads, self.head_dim,
                            self.rotary_emb.cos_sin(qkv.shape[-2], qkv.device))
        # attn.py:110-114: dropout on the attention weights while training
        o = ops.attention(qkv, self.num_heads, self.num_kv_heads, self.head_dim,
                          dropout_p=self.atten_dropout if self.training else 0.0)
        return ops.linear(o, self.o_proj.weight, residual=residual)

    def fused_weight_groups(self):
        """weights read as one matrix by the fused projection: trainer.FlatGradBucket keeps them back to back"""
        return [[self.q_proj.weight, self.k_proj.weight, self.v_proj.weight]]

    @classmethod
    def from_config(cls, input_size: int, hidden_size: int, config: AttentionConfig, positional_embedding: str = "absolute"):
        return cls(input_size=input_size, hidden_size=hidden_size, positional_embedding=positional_embedding,
                   **shallow_asdict(config))


class FFN(nn.Module):
    """SwiGLU: w2(silu(w1 x) * w3 x)  (attn.py:150-156); w1|w3 run as one GEMM with the gate as its epilogue, the residual
    rides w2's epilogue, the gate's gradient rides the epilogue of dY w2 (ops._SwiGLUFFN)."""

    def __init__(self, input_size: int, ffn_hidden_size: int, use_conditional_norm: bool = False, cond_norm_hidden_size: int = 4):
        super().__init__()
        self.w1 = nn.Linear(input_size, ffn_hidden_size, bias=False)
        self.w2 = nn.Linear(ffn_hidden_size, input_size, bias=False)
        self.w3 = nn.Linear(input_size, ffn_hidden_size, bias=False)
        self.correction = ConditionedNorm(1, input_size, cond_norm_hidden_size) if use_conditional_norm else None

    def fused_weight_groups(self):
        return [[self.w1.weight, self.w3.weight]]

    def forward_normed(self, x, norm):
        """norm(x) + ffn(norm(x)) as ONE autograd node (ops._NormedSwiGLUFFN: the norm's gradient kernel sums the K slabs of the
        input-gradient product itself), or None where that form does not apply (conditional norms, CPU, odd shapes)."""
        if self.correction is not None or not x.is_cuda or type(norm) is not RMSNorm:
            return None
        ops.adopt_adjacent_storage([self.w1.weight, self.w3.weight])
        return ops.normed_swiglu_ffn(x, norm.weight, norm.eps, self.w1.weight, self.w3.weight, self.w2.weight)

    def forward(self, x, condition=None, residual=None):
        if x.is_cuda:
            ops.adopt_adjacent_storage([self.w1.weight, self.w3.weight])
        if self.correction is None:
            return ops.swiglu_ffn(x, self.w1.weight, self.w3.weight, self.w2.weight, residual=residual)
        y = self.correction(c=condition, x=ops.swiglu_ffn(x, self.w1.weight, self.w3.weight, self.w2.weight))
        return y if residual is None else residual + y


class TransformerBlock(nn.Module):
    def __init__(self, input_size: int, config: TransformerConfig, skip_connection: bool = False):
        super().__init__()
        hidden = _cfg_get(config, "hidden_size")
        acfg = _cfg_get(config, "attn_config")
        self.attn = GroupQueryFlashAttention.from_config(input_size=input_size, hidden_size=hidden, config=acfg,
                                                         positional_embedding=_cfg_get(config, "positional_embedding"))
        self.ffn = FFN(input_size=input_size, ffn_hidden_size=hidden * _cfg_get(config, "ffn_multiplier"),
                       use_conditional_norm=_cfg_get(acfg, "use_conditional_norm"),
                       cond_norm_hidden_size=_cfg_get(acfg, "cond_norm_hidden_size"))
        eps = _cfg_get(config, "norm_eps")
        self.attn_norm = RMSNorm(input_size, eps=eps) if _cfg_get(config, "use_attn_norm") else None
        self.ffn_norm = RMSNorm(input_size, eps=eps) if _cfg_get(config, "use_ffn_norm") else None
        self.skip_connection = skip_connection
        if skip_connection:
            self.skip_proj = nn.Linear(input_size * 2, input_size)

    def forward(self, x, condition=None, relative_positions=None, skip=None, want_input_alias: bool = False):
        """want_input_alias: also return an alias of the block's INPUT taken from the pre-norm fork (Transformer keeps it as the
        long-range skip): the skip's gradient then joins the stream's inside the norm-gradient kernel instead of by an add launch."""
        if self.skip_connection and skip is not None:          # cat([x, skip]) @ W^T + b as a split-K-operand GEMM
            x = ops.linear(x, self.skip_proj.weight, self.skip_proj.bias, x2=skip)
        alias = x
        if self.attn_norm is None:
            h = x
        elif want_input_alias:
            x, h, alias = ops.rms_norm_fork(x, self.attn_norm.weight, self.attn_norm.eps, with_skip_alias=True)
        else:       # (x, norm(x)) from one node: the residual's gradient is added inside the norm-gradient kernel
            x, h = ops.rms_norm_fork(x, self.attn_norm.weight, self.attn_norm.eps)
        h = self.attn(h, condition=condition, relative_positions=relative_positions, residual=x)   # x + attn(h)
        out = self.ffn.forward_normed(h, self.ffn_norm) if self.ffn_norm is not None else None
        if out is None:
            h = h if self.ffn_norm is None else self.ffn_norm(h)
            out = self.ffn(h, condition=condition, residual=h)     # residual on the NORMALISED stream (attn.py:231-232)
        return (out, alias) if want_input_alias else out


class Transformer(nn.Module):
    def __init__(self, input_size: int, output_size: int, config: TransformerConfig = None):
        super().__init__()
        config = TransformerConfig() if config is None else config
        hidden = _cfg_get(config, "hidden_size")
        n = _cfg_get(config, "num_layers")
        self.use_long_range_skip = _cfg_get(config, "use_long_range_skip")
        if input_size != hidden:
            self.input_proj = nn.Linear(input_size, hidden)
            work = hidden
        else:
            self.input_proj = nn.Identity()
            work = input_size
        self.output_proj = nn.Linear(work, output_size) if work != output_size else nn.Identity()
        self.encoder_layers = nn.ModuleList(TransformerBlock(work, config, False) for _ in range(n // 2))
        self.middle_layer = TransformerBlock(work, config, False) if n % 2 == 1 else None
        self.decoder_layers = nn.ModuleList(TransformerBlock(work, config, True) for _ in range(n // 2))

    def blocks_in_order(self):
        return list(self.encoder_layers) + ([self.middle_layer] if self.middle_layer is not None else []) + list(self.decoder_layers)

    def forward(self, x, condition=None, relative_positions=None):
        if isinstance(self.input_proj, nn.Linear):
            x = ops.linear(x, self.input_proj.weight, self.input_proj.bias)
        skips = []
        left = len(self.blocks_in_order())          # a cut point (staged backward, trainer.TrainStep) after every block but the last
        # the output of encoder layer i is the long-range skip of a decoder layer AND the input of the next block: the skip is taken
        # as an alias handed out by that next block's pre-norm fork (same values; its gradient is added inside the norm-gradient kernel)
        blocks = self.blocks_in_order()
        n_enc = len(self.encoder_layers)
        pending_skip = False                         # the previous block's output is owed to `skips`
        # (a staged backward cuts the graph between blocks: an alias handed out by the NEXT block's node would be reached from two
        # separate backward passes, so with cut points active the skip is the block output itself, as in the reference)
        via_fork = ops._CUT_HOOK[0] is None
        for bi, blk in enumerate(blocks):
            is_dec = bi >= len(blocks) - len(self.decoder_layers)
            if pending_skip and is_dec:              # no block in between: the first decoder layer's skip is its own input
                skips.append(x)
                pending_skip = False
            s = skips.pop() if (is_dec and self.use_long_range_skip) else None
            if pending_skip and not via_fork:
                skips.append(x)
                pending_skip = False
            if pending_skip:                         # this block forks its input first thing: take the skip alias from the fork
                x, alias = blk(x, condition=condition, relative_positions=relative_positions, skip=s, want_input_alias=True)
                skips.append(alias)
            else:
                x = blk(x, condition=condition, relative_positions=relative_positions, skip=s)
            pending_skip = bi < n_enc
            left -= 1
            x = ops.cut(x) if left else x
        if isinstance(self.output_proj, nn.Linear):
            x = ops.linear(x, self.output_proj.weight, self.output_proj.bias)
        return x
