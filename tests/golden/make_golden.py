#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the *imported reference*.

Runs ONLY in the build container (needs /root/reference).  The reference Python never
travels to the GPU box: what is committed is this script plus the .npz arrays it wrote.

The reference hard-imports three packages that are absent here:
  * torch_scatter         -> pure-torch stand-in written below (upstream semantics:
                             empty segment -> 0, mean divides by max(count,1))
  * omegaconf             -> empty stub (only imported, never used on this path)
  * rotary_embedding_torch-> stand-in restating the published RotaryEmbedding(dim) defaults (freqs_for='lang', theta=1e4,
                             learned_freq=False: `freqs` is a frozen nn.Parameter; rotate_queries_or_keys rotates the
                             interleaved pairs (2i, 2i+1) by position * freqs[i]); used by the `rope` case only
The stand-ins are written to a temp dir, never into the repo tree.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import os
import sys
import tempfile
import textwrap
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

SCATTER_STANDIN = textwrap.dedent('''
    import torch
    def _ids(ip):
        deg = ip[1:] - ip[:-1]
        return torch.repeat_interleave(torch.arange(deg.numel(), device=deg.device), deg), deg
    def segment_csr(src, indptr, out=None, reduce="sum"):
        dim = indptr.dim() - 1
        ids, deg = _ids(indptr.reshape(-1, indptr.shape[-1])[0])
        shape = list(src.shape); shape[dim] = deg.numel()
        ish = [1] * src.dim(); ish[dim] = -1
        index = ids.view(ish).expand_as(src)
        res = torch.zeros(shape, dtype=src.dtype, device=src.device)
        if reduce in ("sum", "mean"):
            res.scatter_add_(dim, index, src)
            if reduce == "mean":
                res = res / deg.clamp(min=1).to(src.dtype).view(ish)
        elif reduce == "max":
            res.scatter_reduce_(dim, index, src, "amax", include_self=False)
        else:
            raise ValueError(reduce)
        return res
    def scatter_sum(src, index, dim=0, out=None, dim_size=None):
        shape = list(src.shape); shape[dim] = dim_size
        return torch.zeros(shape, dtype=src.dtype, device=src.device).index_add_(dim, index, src)
    def scatter_mean(src, index, dim=0, out=None, dim_size=None):
        cnt = torch.bincount(index, minlength=dim_size).clamp(min=1).to(src.dtype)
        sh = [1] * src.dim(); sh[dim] = -1
        return scatter_sum(src, index, dim, None, dim_size) / cnt.view(sh)
    def scatter_max(src, index, dim=0, out=None, dim_size=None):
        shape = list(src.shape); shape[dim] = dim_size
        sh = [1] * src.dim(); sh[dim] = -1
        res = torch.zeros(shape, dtype=src.dtype, device=src.device)
        res.scatter_reduce_(dim, index.view(sh).expand_as(src), src, "amax", include_self=False)
        return res, None
''')


ROTARY_STANDIN = textwrap.dedent('''
    import torch
    from torch import nn
    class RotaryEmbedding(nn.Module):
        def __init__(self, dim, theta=10000):
            super().__init__()
            freqs = 1. / (theta ** (torch.arange(0, dim, 2)[:(dim // 2)].float() / dim))
            self.freqs = nn.Parameter(freqs, requires_grad=False)
        def rotate_queries_or_keys(self, t, seq_dim=-2):
            seq_len = t.shape[seq_dim]
            seq = torch.arange(seq_len, device=t.device, dtype=t.dtype)
            freqs = torch.einsum('..., f -> ... f', seq, self.freqs.to(t.dtype))
            freqs = freqs.repeat_interleave(2, dim=-1)                     # repeat '... n -> ... (n r)', r = 2
            x = t.reshape(*t.shape[:-1], -1, 2)
            x1, x2 = x.unbind(dim=-1)
            rot = torch.stack((-x2, x1), dim=-1).reshape(t.shape)          # rotate_half
            return t * freqs.cos() + rot * freqs.sin()
''')


def install_standins():
    root = tempfile.mkdtemp(prefix="gaot_ref_standins_")
    for pkg in ("torch_scatter", "omegaconf", "rotary_embedding_torch"):
        os.makedirs(os.path.join(root, pkg))
    with open(os.path.join(root, "torch_scatter", "segment_csr.py"), "w") as f:
        f.write(SCATTER_STANDIN)
    with open(os.path.join(root, "torch_scatter", "__init__.py"), "w") as f:
        f.write("from .segment_csr import segment_csr, scatter_sum, scatter_mean, scatter_max\n")
    with open(os.path.join(root, "omegaconf", "__init__.py"), "w") as f:
        f.write("class DictConfig(dict): pass\nclass OmegaConf: pass\n")
    with open(os.path.join(root, "rotary_embedding_torch", "__init__.py"), "w") as f:
        f.write(ROTARY_STANDIN)
    sys.path.insert(0, root)
    sys.path.insert(1, REF)


def grid(sizes):
    axes = [torch.linspace(-1, 1, n) for n in sizes]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), -1).reshape(-1, len(sizes))


def to_np(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        else:
            out[k] = np.asarray(v)
    return out


# Every case: magno kwargs, transformer kwargs, data shapes.
BASE_M = dict(coord_dim=2, radius=0.2, hidden_size=8, mlp_layers=3, lifting_channels=8,
              neighbor_search_method="native")
BASE_T = dict(patch_size=2, hidden_size=32)
BASE_A = dict(num_heads=4, num_kv_heads=4)

CASES = {
    # name: (magno overrides, transformer overrides, attn overrides, data spec)
    "fx2d_base":      ({}, {}, {}, dict(N=256, lat=[16, 16], B=2, cin=2, cout=1, seed=0, full=True)),
    "fx2d_base_s1":   ({}, {}, {}, dict(N=200, lat=[16, 16], B=3, cin=1, cout=2, seed=1)),
    "fx2d_zero_deg":  ({"radius": 0.12}, {}, {}, dict(N=64, lat=[16, 16], B=2, cin=1, cout=1, seed=0)),
    "fx2d_headdim32": ({"lifting_channels": 16, "hidden_size": 16}, {"hidden_size": 64}, {"num_heads": 2, "num_kv_heads": 2},
                       dict(N=300, lat=[16, 16], B=2, cin=1, cout=1, seed=2, light=True)),
    "fx2d_inproj":    ({}, {"hidden_size": 48}, {"num_heads": 4, "num_kv_heads": 2},
                       dict(N=256, lat=[16, 16], B=2, cin=1, cout=1, seed=3, light=True)),
    "vx2d":           ({"precompute_edges": True}, {}, {}, dict(N=180, lat=[16, 16], B=3, cin=3, cout=1, seed=0, vx=True)),
    "ms_mean":        ({"scales": [1.0, 2.0], "radius": 0.12}, {}, {}, dict(N=256, lat=[16, 16], B=2, cin=1, cout=1, seed=0)),
    "ms_weighted":    ({"scales": [1.0, 2.0], "radius": 0.12, "use_scale_weights": True}, {}, {},
                       dict(N=256, lat=[16, 16], B=2, cin=1, cout=1, seed=1)),
    "attn_dot":       ({"attention_type": "dot_product"}, {}, {}, dict(N=256, lat=[16, 16], B=2, cin=1, cout=1, seed=0)),
    "no_attn_mean":   ({"use_attention": False}, {}, {}, dict(N=256, lat=[16, 16], B=2, cin=1, cout=1, seed=0)),
    "no_geoembed":    ({"use_geoembed": False}, {}, {}, dict(N=256, lat=[16, 16], B=2, cin=1, cout=1, seed=0)),
    "nonlinear":      ({"transform_type": "nonlinear"}, {}, {}, dict(N=256, lat=[16, 16], B=2, cin=8, cout=1, seed=0)),
    "node_embed":     ({"node_embedding": True}, {}, {}, dict(N=256, lat=[16, 16], B=2, cin=1, cout=1, seed=0, light=True)),
    "pointnet":       ({"embedding_method": "pointnet"}, {}, {}, dict(N=256, lat=[16, 16], B=2, cin=1, cout=1, seed=0, light=True)),
    "fx3d":           ({"coord_dim": 3, "radius": 0.45, "lifting_channels": 6}, {"hidden_size": 48}, {},
                       dict(N=300, lat=[8, 8, 8], B=2, cin=3, cout=1, seed=0)),
    "even_layers":    ({}, {"num_layers": 4}, {}, dict(N=256, lat=[16, 16], B=2, cin=1, cout=1, seed=4, light=True)),
    # round 2
    "linear_kernelonly":    ({"transform_type": "linear_kernelonly"}, {}, {}, dict(N=256, lat=[16, 16], B=2, cin=2, cout=1, seed=5)),
    "nonlinear_kernelonly": ({"transform_type": "nonlinear_kernelonly"}, {}, {}, dict(N=256, lat=[16, 16], B=2, cin=8, cout=1, seed=6)),
    "rope":           ({}, {"positional_embedding": "rope"}, {"num_heads": 4, "num_kv_heads": 2},
                       dict(N=256, lat=[16, 16], B=2, cin=1, cout=1, seed=7)),
    "pointnet_mean":  ({"embedding_method": "pointnet", "pooling": "mean"}, {}, {}, dict(N=256, lat=[16, 16], B=2, cin=1, cout=1, seed=8, light=True)),
}


def build(case):
    from src.model.gaot import GAOT
    from src.model.layers.magno import MAGNOConfig
    from src.model.layers.attn import TransformerConfig, AttentionConfig
    mo, to, ao, ds = CASES[case]
    m = dict(BASE_M); m.update(mo)
    t = dict(BASE_T); t.update(to)
    a = dict(BASE_A); a.update(ao)
    torch.manual_seed(1000 + ds["seed"])
    cfg = NS(args=NS(magno=MAGNOConfig(**m), transformer=TransformerConfig(attn_config=AttentionConfig(**a), **t)),
             latent_tokens_size=ds["lat"])
    model = GAOT(ds["cin"], ds["cout"], cfg)
    return model, m, t, a, ds


def run_case(case):
    from src.model.layers.utils.neighbor_search import NeighborSearch
    model, m, t, a, ds = build(case)
    d = m["coord_dim"]
    g = torch.Generator().manual_seed(ds["seed"])
    lat = grid(ds["lat"])
    B, N = ds["B"], ds["N"]
    vx = ds.get("vx", False)
    x = (torch.rand(B, N, d, generator=g) if vx else torch.rand(N, d, generator=g)) * 2 - 1
    p = torch.randn(B, N, ds["cin"], generator=g)
    tgt = torch.randn(B, N, ds["cout"], generator=g)
    out = {"meta.case": case, "meta.magno": repr(m), "meta.transformer": repr(t), "meta.attn": repr(a),
           "in.latent": lat, "in.xcoord": x, "in.pndata": p, "in.target": tgt}
    kwargs = dict(latent_tokens_coord=lat, xcoord=x, pndata=p)
    scales = m.get("scales", [1.0])
    if vx:
        ns = NeighborSearch("native")
        enc = [[ns(x[b], lat, m["radius"] * s) for s in scales] for b in range(B)]
        dec = [[ns(lat, x[b], m["radius"] * s) for s in scales] for b in range(B)]
        kwargs.update(encoder_nbrs=enc, decoder_nbrs=dec)
        for b in range(B):
            for si in range(len(scales)):
                out[f"csr.enc.b{b}.s{si}.index"] = enc[b][si]["neighbors_index"]
                out[f"csr.enc.b{b}.s{si}.splits"] = enc[b][si]["neighbors_row_splits"]
                out[f"csr.dec.b{b}.s{si}.index"] = dec[b][si]["neighbors_index"]
                out[f"csr.dec.b{b}.s{si}.splits"] = dec[b][si]["neighbors_row_splits"]
    for k, v in model.state_dict().items():
        out[f"w.{k}"] = v.clone()

    # ---- intermediates through forward hooks on the reference modules ----
    inter = {}

    def hook(name):
        def fn(mod, args, res):
            inter.setdefault(name, []).append(res.detach().clone())
        return fn

    hs = []
    if not vx:
        hs.append(model.encoder.agno.register_forward_hook(hook("enc.agno")))
        hs.append(model.decoder.agno.register_forward_hook(hook("dec.agno")))
        if m.get("use_geoembed", True):
            hs.append(model.encoder.geoembed.register_forward_hook(hook("enc.geoembed")))
            hs.append(model.decoder.geoembed.register_forward_hook(hook("dec.geoembed")))
    hs.append(model.encoder.register_forward_hook(hook("enc.out")))
    hs.append(model.processor.register_forward_hook(hook("proc.transformer_out")))
    hs.append(model.patch_linear.register_forward_hook(hook("proc.patch_linear")))
    for nm, blk in list(model.processor.encoder_layers.named_children()):
        hs.append(blk.register_forward_hook(hook(f"proc.enc{nm}")))
    if model.processor.middle_layer is not None:
        hs.append(model.processor.middle_layer.register_forward_hook(hook("proc.mid")))
    for nm, blk in list(model.processor.decoder_layers.named_children()):
        hs.append(blk.register_forward_hook(hook(f"proc.dec{nm}")))

    model.train()
    pred = model(**kwargs)
    for h in hs:
        h.remove()
    for k, lst in inter.items():
        for i, v in enumerate(lst):
            out[f"mid.{k}.{i}" if len(lst) > 1 else f"mid.{k}"] = v
    out["out.pred"] = pred.detach().clone()

    if not vx:  # cached CSR + geometry-only intermediates, straight from the reference's own methods
        encn = list(model.encoder.neighbor_cache.values())[0]
        decn = list(model.decoder.neighbor_cache.values())[0]
        for si in range(len(scales)):
            out[f"csr.enc.s{si}.index"] = encn[si]["neighbors_index"]
            out[f"csr.enc.s{si}.splits"] = encn[si]["neighbors_row_splits"]
            out[f"csr.dec.s{si}.index"] = decn[si]["neighbors_index"]
            out[f"csr.dec.s{si}.splits"] = decn[si]["neighbors_row_splits"]
        if m.get("use_geoembed", True) and m.get("embedding_method", "statistical") == "statistical":
            out["mid.enc.geo_stats"] = model.encoder.geoembed._compute_statistical_features(x, lat, encn[0])
            out["mid.dec.geo_stats"] = model.decoder.geoembed._compute_statistical_features(lat, x, decn[0])
        if m.get("use_attention", True) and m.get("attention_type", "cosine") == "cosine" and not m.get("node_embedding", False):
            import torch.nn.functional as F
            idx, sp = encn[0]["neighbors_index"], encn[0]["neighbors_row_splits"]
            rep = torch.repeat_interleave(lat, sp[1:] - sp[:-1], dim=0)
            sc = torch.sum(F.normalize(rep, dim=-1) * F.normalize(x[idx], dim=-1), dim=-1)
            out["mid.enc.attn"] = model.encoder.agno._segment_softmax(sc, sp)
            if m.get("transform_type", "linear") == "linear":
                out["mid.enc.kernel"] = model.encoder.agno.channel_mlp(torch.cat([x[idx], rep], dim=-1))

    light = ds.get("light", False)
    # ---- one trainer step: MSE -> backward -> AdamW (static_trainer.py:160-178, optimizers.py:196,247-257)
    loss = torch.nn.MSELoss()(pred, tgt)
    opt = torch.optim.AdamW(model.parameters(), lr=8e-4, weight_decay=1e-5)
    opt.zero_grad()
    loss.backward()
    out["out.loss"] = loss.detach().clone()
    for k, prm in model.named_parameters():
        gk = prm.grad if prm.grad is not None else torch.zeros_like(prm)
        if light:
            out[f"gnorm.{k}"] = gk.norm()
        else:
            out[f"g.{k}"] = gk.clone()
    if ds.get("full", False):
        opt.step()
        for k, prm in model.named_parameters():
            out[f"w1.{k}"] = prm.detach().clone()
    np.savez_compressed(os.path.join(HERE, f"{case}.npz"), **to_np(out))
    print(f"{case}: pred {tuple(pred.shape)} loss {float(loss):.6f}")


def run_condnorm_rollout():
    """Sequential model with cond-norm: one pair-training forward + 3-step rollouts in all stepper modes."""
    from src.model.gaot import GAOT
    from src.model.layers.magno import MAGNOConfig
    from src.model.layers.attn import TransformerConfig, AttentionConfig
    torch.manual_seed(77)
    m = dict(BASE_M)
    t = dict(BASE_T)
    a = dict(BASE_A, use_conditional_norm=True)
    cfg = NS(args=NS(magno=MAGNOConfig(**m), transformer=TransformerConfig(attn_config=AttentionConfig(**a), **t)),
             latent_tokens_size=[16, 16])
    udim, cdim = 2, 1
    model = GAOT(udim + cdim + 1, udim, cfg)   # cond-norm drops the last (dt) column: sequential_trainer.py:192-198
    g = torch.Generator().manual_seed(5)
    lat = grid([16, 16])
    N, B = 220, 2
    x = torch.rand(N, 2, generator=g) * 2 - 1
    xb = torch.randn(B, N, udim + cdim + 2, generator=g)
    stats = {"u": {"mean": torch.tensor([0.1, -0.2]), "std": torch.tensor([1.5, 0.7])},
             "c": {"mean": torch.tensor([0.0]), "std": torch.tensor([1.0])},
             "res": {"mean": torch.tensor([0.01, 0.02]), "std": torch.tensor([0.3, 0.4])},
             "der": {"mean": torch.tensor([-0.05, 0.03]), "std": torch.tensor([0.8, 1.1])},
             "start_time": {"mean": 0.4, "std": 0.25}, "time_diffs": {"mean": 0.2, "std": 0.1}}
    t_values = np.linspace(0.0, 1.0, 11)
    time_indices = np.array([0, 2, 4, 6])
    out = {"meta.case": "condnorm_rollout", "meta.magno": repr(m), "meta.transformer": repr(t), "meta.attn": repr(a),
           "in.latent": lat, "in.xcoord": x, "in.x_batch": xb, "in.t_values": t_values, "in.time_indices": time_indices}
    for k, v in model.state_dict().items():
        out[f"w.{k}"] = v.clone()
    for grp in ("u", "c", "res", "der"):
        out[f"stats.{grp}.mean"] = stats[grp]["mean"]
        out[f"stats.{grp}.std"] = stats[grp]["std"]
    for grp in ("start_time", "time_diffs"):
        out[f"stats.{grp}.mean"] = stats[grp]["mean"]
        out[f"stats.{grp}.std"] = stats[grp]["std"]
    model.eval()
    with torch.no_grad():
        out["out.pair_forward"] = model(latent_tokens_coord=lat, xcoord=x, pndata=xb[..., :-1], condition=xb[..., 0, -2:-1])
    for mode in ("output", "residual", "time_der"):
        out[f"out.rollout.{mode}"] = model.autoregressive_predict(
            x_batch=xb[..., :udim + cdim], time_indices=time_indices, t_values=t_values, stats=stats,
            stepper_mode=mode, latent_tokens_coord=lat, fixed_coord=x, use_conditional_norm=True)
    np.savez_compressed(os.path.join(HERE, "condnorm_rollout.npz"), **to_np(out))
    print("condnorm_rollout:", tuple(out["out.rollout.output"].shape))



def run_condnorm_train():
    """Sequential model with cond-norm, ONE TRAINING STEP the way the reference's sequential trainer takes it
    (sequential_trainer.py:182-204: pndata = x_batch[..., :-1], condition = x_batch[..., 0, -2:-1]; MSE -> backward -> AdamW,
    optimizers.py:247-257): loss, every parameter gradient -- the `correction.mlp_{scale,bias}` parameters of every attention and
    FFN block included (mlp.py:74-124, attn.py:89-90,150-156) -- and the weights after the update.  Same model and data recipe
    as run_condnorm_rollout (seed 77 / generator 5) plus a seeded target."""
    from src.model.gaot import GAOT
    from src.model.layers.magno import MAGNOConfig
    from src.model.layers.attn import TransformerConfig, AttentionConfig
    torch.manual_seed(77)
    m = dict(BASE_M)
    t = dict(BASE_T)
    a = dict(BASE_A, use_conditional_norm=True)
    cfg = NS(args=NS(magno=MAGNOConfig(**m), transformer=TransformerConfig(attn_config=AttentionConfig(**a), **t)),
             latent_tokens_size=[16, 16])
    udim, cdim = 2, 1
    model = GAOT(udim + cdim + 1, udim, cfg)
    g = torch.Generator().manual_seed(5)
    lat = grid([16, 16])
    N, B = 220, 2
    x = torch.rand(N, 2, generator=g) * 2 - 1
    xb = torch.randn(B, N, udim + cdim + 2, generator=g)
    tgt = torch.randn(B, N, udim, generator=torch.Generator().manual_seed(6))
    out = {"meta.case": "condnorm_train", "meta.magno": repr(m), "meta.transformer": repr(t), "meta.attn": repr(a),
           "in.latent": lat, "in.xcoord": x, "in.x_batch": xb, "in.target": tgt}
    for k, v in model.state_dict().items():
        out[f"w.{k}"] = v.clone()
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=8e-4, weight_decay=1e-5)
    for step in range(2):          # two steps: the second runs on weights whose correction MLPs have moved
        opt.zero_grad()
        pred = model(latent_tokens_coord=lat, xcoord=x, pndata=xb[..., :-1], condition=xb[..., 0, -2:-1])
        loss = torch.nn.MSELoss()(pred, tgt)
        loss.backward()
        if step == 0:
            out["out.pred"] = pred.detach().clone()
        out[f"out.loss{step}"] = loss.detach().clone()
        for k, prm in model.named_parameters():
            out[f"g{step}.{k}"] = (prm.grad if prm.grad is not None else torch.zeros_like(prm)).clone()
        opt.step()
        for k, prm in model.named_parameters():
            out[f"w{step + 1}.{k}"] = prm.detach().clone()
    encn = list(model.encoder.neighbor_cache.values())[0]
    decn = list(model.decoder.neighbor_cache.values())[0]
    out["csr.enc.s0.index"], out["csr.enc.s0.splits"] = encn[0]["neighbors_index"], encn[0]["neighbors_row_splits"]
    out["csr.dec.s0.index"], out["csr.dec.s0.splits"] = decn[0]["neighbors_index"], decn[0]["neighbors_row_splits"]
    np.savez_compressed(os.path.join(HERE, "condnorm_train.npz"), **to_np(out))
    corr = [k for k in out if k.startswith("g0.") and "correction" in k]
    print(f"condnorm_train: loss {float(out['out.loss0']):.6f} -> {float(out['out.loss1']):.6f}; {len(corr)} correction gradient tensors, "
          f"largest {max(float(out[k].abs().max()) for k in corr):.3e}")


def run_c2_stats_gates():
    """The reference's OWN float32 -> float64 movement at the bench configuration (BASELINE configs[1]: 16 384 nodes, batch 8, the
    example model; weights and data exactly as bench.py builds them).

    Four gradient tensors sit right behind ReLU gates that read the standardised geometry statistics (gemb.py:103-171, 54-59):
    with 16 384 x 64 gates a few pre-activations lie within fp32 rounding of zero, and which side they fall on depends on the
    arithmetic the statistics were computed in.  This fixture pins that statement to the reference itself: the same model, weights
    and batch are run in float32 and in float64 (`model.type(torch.float64)`, base_trainer.py:173-179).  NOTE: the reference's
    float64 path raises at gemb.py:153 (`PCA_features = torch.zeros(...)` is float32 whatever the model dtype); it only runs with
    torch's DEFAULT dtype switched to float64 around the call, which is what is done here -- the reference sources are untouched.
    Stored: both losses, the relative movement of the prediction and of EVERY gradient tensor, the full float32 and float64
    gradients of the four gated tensors, and checksums of the initial weights (the GPU test rebuilds them from the seed)."""
    import bench
    from src.model.gaot import GAOT
    from src.model.layers.magno import MAGNOConfig
    from src.model.layers.attn import TransformerConfig
    from oracle import gaot_oracle as O
    torch.manual_seed(0)
    mcfg = MAGNOConfig(coord_dim=2, radius=bench.RADIUS, hidden_size=64, mlp_layers=3, lifting_channels=bench.C_LIFT,
                       neighbor_search_method="grid", precompute_edges=True)
    tcfg = TransformerConfig(patch_size=bench.PATCH, hidden_size=bench.HIDDEN)
    model = GAOT(1, 1, NS(args=NS(magno=mcfg, transformer=tcfg), latent_tokens_size=bench.LATENT))
    lat, x, p, t = bench.synthetic(1234, torch.device("cpu"))
    enc, dec = O.radius_csr(x, lat, bench.RADIUS, exact=True), O.radius_csr(lat, x, bench.RADIUS, exact=True)      # = the `grid` backend's lists
    nb = lambda c: {"neighbors_index": c[0], "neighbors_row_splits": c[1]}
    out = {"meta.case": "c2_stats_gates", "meta.note": "reference GAOT at the bench configuration, float32 and float64 on identical weights"}
    for k, v in model.state_dict().items():
        out[f"wsum.{k}"] = np.array([float(v.double().sum()), float(v.double().norm())])
    res = {}
    for dt in (torch.float32, torch.float64):
        torch.set_default_dtype(dt)
        try:
            m = model.type(dt).train()
            m.zero_grad(set_to_none=True)
            pred = m(latent_tokens_coord=lat.to(dt), xcoord=x.to(dt), pndata=p.to(dt), encoder_nbrs=[nb(enc)], decoder_nbrs=[nb(dec)])
            loss = torch.nn.MSELoss()(pred, t.to(dt))
            loss.backward()
        finally:
            torch.set_default_dtype(torch.float32)
        res[dt] = ({k: q.grad.detach().double().clone() for k, q in m.named_parameters()}, pred.detach().double(), float(loss.detach()))
    (g32, p32, l32), (g64, p64, l64) = res[torch.float32], res[torch.float64]
    top = max(float(v.norm()) for v in g64.values())
    out["loss32"], out["loss64"] = np.float64(l32), np.float64(l64)
    out["pred_rel_move"] = np.float64(float((p32 - p64).norm() / p64.norm()))
    out["grad_norm_top"] = np.float64(top)
    gated = ("encoder.geoembed.mlp.0.weight", "encoder.geoembed.mlp.0.bias", "decoder.geoembed.mlp.0.weight", "decoder.geoembed.mlp.0.bias")
    for k in g64:
        out[f"move.{k}"] = np.float64(float((g32[k] - g64[k]).norm()) / max(float(g64[k].norm()), 1e-3 * top))
        out[f"gnorm64.{k}"] = np.float64(float(g64[k].norm()))
    for k in gated:
        out[f"g32.{k}"] = g32[k].float()
        out[f"g64.{k}"] = g64[k]                  # float64
    np.savez_compressed(os.path.join(HERE, "c2_stats_gates.npz"), **to_np(out))
    worst = sorted(((float(out[f"move.{k}"]), k) for k in g64), reverse=True)[:4]
    print("c2_stats_gates: loss32 %.9f loss64 %.9f; prediction moves %.2e; gradients: %s" % (l32, l64, float(out["pred_rel_move"]),
          ", ".join(f"{k} {v:.2e}" for v, k in worst)))


def run_neighbor_kats():
    """CSR known answers from the reference's in-repo backends, incl. points exactly at distance r."""
    from src.model.layers.utils.neighbor_search import NeighborSearch
    out = {}
    g = torch.Generator().manual_seed(11)
    # (a) exact-boundary: integer lattice, radius 1.0 -> axis neighbours are at distance exactly r (inclusive <=)
    lattice = torch.stack(torch.meshgrid(torch.arange(5.), torch.arange(5.), indexing="ij"), -1).reshape(-1, 2)
    qs = torch.tensor([[2., 2.], [0., 0.], [4., 1.], [10., 10.], [2.5, 2.5]])
    for meth in ("native", "chunked", "grid"):
        r = NeighborSearch(meth)(lattice, qs, 1.0)
        out[f"lattice.{meth}.index"] = r["neighbors_index"]
        out[f"lattice.{meth}.splits"] = r["neighbors_row_splits"]
    out["lattice.data"], out["lattice.queries"], out["lattice.radius"] = lattice, qs, 1.0
    # (b) random 2-D and 3-D
    for d, n, mq, rad in ((2, 400, 144, 0.15), (3, 500, 125, 0.4)):
        data = torch.rand(n, d, generator=g) * 2 - 1
        q = grid([12, 12] if d == 2 else [5, 5, 5])
        for meth in ("native", "chunked") + (("grid",) if d == 2 else ()):
            r = NeighborSearch(meth)(data, q, rad)
            out[f"rand{d}d.{meth}.index"] = r["neighbors_index"]
            out[f"rand{d}d.{meth}.splits"] = r["neighbors_row_splits"]
        out[f"rand{d}d.data"], out[f"rand{d}d.queries"], out[f"rand{d}d.radius"] = data, q, rad
    np.savez_compressed(os.path.join(HERE, "neighbor_kats.npz"), **to_np(out))
    print("neighbor_kats: ok")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs the reference checkout at /root/reference (build container only)")
    install_standins()
    torch.set_num_threads(8)
    only = sys.argv[1:]
    for c in CASES:
        if not only or c in only:
            run_case(c)
    if not only or "condnorm_rollout" in only:
        run_condnorm_rollout()
    if not only or "condnorm_train" in only:
        run_condnorm_train()
    if not only or "neighbor_kats" in only:
        run_neighbor_kats()
    if not only or "c2_stats_gates" in only:
        sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))      # bench.py (the bench configuration's builders) and the oracle's CSR
        run_c2_stats_gates()
