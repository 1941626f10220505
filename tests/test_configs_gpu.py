"""-m gpu: every BASELINE.json configuration as a workload, HIP path vs the CPU oracle on identical weights and inputs.

  C2  the bench configuration itself: 16 384 uniform nodes, batch 8, radius 0.033, the reference's example model
      (forward rel-L2, loss, every gradient per tensor, post-AdamW weights; default kernels and the fp32-only kernel modes)
  C3  NACA0012-shaped degree-skewed meshes, vx mode (a different mesh per sample), 3 input channels, max encoder degree
      > 256 next to thousands of empty latent rows; also with training-time neighbour sub-sampling on
  C1  1 024-node meshes, batch 4: 43 % of the latent tokens have no neighbour
  C5  3-D surface cloud, 32^3 latent grid = 4 096 tokens of width 384, head_dim 48
  C4  NS-Gauss-shaped time-dependent run: 16 384 nodes, batch 4, in = u(2) + 2 time columns, out 2: a pair-training step and a
      10-step autoregressive rollout (stepper 'time_der'; hipGraph-replayed forward, fused stepper kernels) against the oracle

Tolerances (north_star: <= 1e-5 relative output error, fp32): output rel-L2 <= 1e-5, loss <= 1e-5 relative, every gradient
tensor rel-L2 <= 1e-4 WITHOUT a floor on the denominator: a tensor at or below its own fp32 rounding -- e.g. the key bias under a
softmax, zero in exact arithmetic -- is held to 3x the distance the reference's own fp32 arithmetic keeps from float64 on that
tensor (one float64 oracle pass per test: tests/_golden.py `fp32_noise` / `unfloored_ratio`).
Radius graphs: the oracle's `exact=True` search (explicit differences, the reference's `grid` backend = method 'auto') is the
graph the HIP cell list must reproduce bit for bit; the cdist-based `native` backend differs from it only in pairs within
rounding of the radius (checked below), so numeric parity is always taken on ONE graph.
"""
from types import SimpleNamespace as NS

import pytest
import torch

from tests._golden import fp32_noise, rel_l2, unfloored_ratio
from tests._workloads import grid, naca_points, shell_points, uniform_points

pytestmark = pytest.mark.gpu

OUT_TOL, LOSS_TOL, GRAD_TOL = 1e-5, 1e-5, 1e-4


def dev():
    return torch.device("cuda:0")


def make_model(cin, cout, lat_sizes, d=2, C=64, hidden=256, heads=8, radius=0.033, P=2, precompute=True, seed=0, **magno_kw):
    from gaot_amd.model.gaot import GAOT
    from gaot_amd.model.layers.magno import MAGNOConfig
    from gaot_amd.model.layers.attn import TransformerConfig, AttentionConfig
    from oracle import gaot_oracle as O
    torch.manual_seed(seed)
    mcfg = MAGNOConfig(coord_dim=d, radius=radius, hidden_size=64, mlp_layers=3, lifting_channels=C, precompute_edges=precompute, **magno_kw)
    tcfg = TransformerConfig(patch_size=P, hidden_size=hidden, attn_config=AttentionConfig(num_heads=heads, num_kv_heads=heads))
    model = GAOT(cin, cout, NS(args=NS(magno=mcfg, transformer=tcfg), latent_tokens_size=lat_sizes))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    ocfg = O.OracleConfig(coord_dim=d, radius=radius, hidden_size=64, lifting_channels=C, patch_size=P, tf_hidden_size=hidden,
                          num_heads=heads, num_kv_heads=heads, latent_tokens_size=lat_sizes, precompute_edges=True)
    return model, sd, ocfg


def csr_dict(c):
    return {"neighbors_index": c[0].to(dev()), "neighbors_row_splits": c[1].to(dev())}


def grad_errors(model, grads_ref, noise=None):
    """per-tensor error of the gradients, NO floor on the denominator (tests/_golden.py `unfloored_ratio`): with `noise` = the reference's own
    fp32 rounding per tensor (`fp32_noise`) a tensor's figure is GRAD_TOL x (its error / its bar), the bar being max(GRAD_TOL x its norm,
    3 x its fp32 rounding): for every tensor larger than its rounding that IS its relative L2 error; `< GRAD_TOL` passes.
    noise=None (comparisons against a float64 evaluation that carry their own per-tensor bar): plain relative L2, floored as in rounds 1-5."""
    if noise is not None:
        got = {k: prm.grad for k, prm in model.named_parameters()}
        return {k: GRAD_TOL * v for k, v in unfloored_ratio(got, {k: grads_ref[k] for k in got}, noise, GRAD_TOL).items()}
    top = max(float(g.double().norm()) for g in grads_ref.values())
    out = {}
    for k, prm in model.named_parameters():
        got = prm.grad.detach().cpu().double() if prm.grad is not None else torch.zeros_like(prm).cpu().double()
        ref = grads_ref[k].double()
        out[k] = float((got - ref).norm()) / max(float(ref.norm()), 1e-3 * top)
    return out


def check_step(model, oracle_out, fwd_kwargs, p, tgt, what="", noise=None):
    """one eager forward + MSE + backward of the HIP path against (loss, grads, pred) of the oracle; `noise` = fp32_noise(...) of the same step"""
    assert noise is not None, "check_step: pass noise=fp32_noise(sd, ocfg, batch, grads) (gradients are compared without a floor)"
    from gaot_amd import ops
    loss_ref, grads_ref, pred_ref = oracle_out
    model.zero_grad(set_to_none=True)
    pred = model(pndata=p, **fwd_kwargs)
    loss = ops.mse_loss(pred, tgt)
    loss.backward()
    torch.cuda.synchronize()
    e_out = rel_l2(pred.detach().cpu(), pred_ref)
    e_loss = abs(float(loss.detach()) - float(loss_ref)) / abs(float(loss_ref))
    errs = grad_errors(model, grads_ref, noise)
    worst = max(errs, key=errs.get)
    print(f"[{what}] out rel-L2 {e_out:.2e}  loss rel {e_loss:.2e}  worst grad rel-L2 {errs[worst]:.2e} ({worst})")
    assert e_out < OUT_TOL, (what, e_out)
    assert e_loss < LOSS_TOL, (what, e_loss)
    assert errs[worst] < GRAD_TOL, (what, worst, errs[worst])
    return e_out, e_loss, errs[worst]


# ------------------------------------------------------------------------------------------------ C2: the bench configuration
@pytest.fixture(scope="module")
def c2():
    """bench.py's own model / data builders (BASELINE configs[1]) + ONE oracle train step on them"""
    import bench
    from oracle import gaot_oracle as O
    torch.manual_seed(0)
    model = bench.build_model()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    lat, x, p, t = bench.synthetic(1234, torch.device("cpu"))
    assert x.shape == (16384, 2) and p.shape == (8, 16384, 1) and lat.shape == (4096, 2)
    ocfg = O.OracleConfig(radius=bench.RADIUS, hidden_size=64, lifting_channels=bench.C_LIFT, patch_size=bench.PATCH,
                          tf_hidden_size=bench.HIDDEN, latent_tokens_size=bench.LATENT, precompute_edges=True)
    enc, dec = [O.radius_csr(x, lat, bench.RADIUS, exact=True)], [O.radius_csr(lat, x, bench.RADIUS, exact=True)]
    batch = dict(latent=lat, xcoord=x, pndata=p, target=t, encoder_nbrs=enc, decoder_nbrs=dec)
    loss, grads, new_sd, _, pred = O.train_step(sd, ocfg, batch, lr=8e-4, weight_decay=1e-5, return_pred=True)
    # the same step with the geometry statistics evaluated in float64 (test instrument, see OracleConfig.stats_dtype)
    ocfg64 = O.OracleConfig(**{**ocfg.__dict__, "stats_dtype": "float64"})
    loss64, grads64, new_sd64, _, pred64 = O.train_step(sd, ocfg64, batch, lr=8e-4, weight_decay=1e-5, return_pred=True)
    # ... and the oracle evaluated in float64 END TO END (weights, batch, every intermediate): the measuring stick that separates the
    # kernels' rounding from the fp32 reference's own
    dbl = lambda v: v.double() if torch.is_tensor(v) and v.is_floating_point() else v
    _, gradsd, _, _, predd = O.train_step({k: dbl(v) for k, v in sd.items()}, ocfg, {k: dbl(v) for k, v in batch.items()}, return_pred=True)
    noise = {k: float((grads[k].double() - gradsd[k]).norm()) for k in gradsd}          # the reference's own fp32 rounding per tensor
    noise64 = {k: max(noise[k], float((grads64[k].double() - gradsd[k]).norm())) for k in gradsd}
    return NS(sd=sd, lat=lat, x=x, p=p, t=t, enc=enc, dec=dec, loss=loss, grads=grads, new_sd=new_sd, pred=pred,
              loss64=loss64, grads64=grads64, new_sd64=new_sd64, pred64=pred64, gradsd=gradsd, predd=predd, noise=noise, noise64=noise64)


def _c2_model(c2):
    import bench
    m = bench.build_model()
    m.load_state_dict(c2.sd)
    return m.to(dev()).train()


# gradient tensors right behind the ReLU gates that read the geometry statistics: their value moves by ~2e-4 (relative L2)
# when the ORACLE ITSELF evaluates those statistics in float64 instead of float32 (16 384 x 64 gates, a few of them within
# rounding of zero flip) -- the reference's own conditioning.  The HIP path computes the statistics in float64.
STATS_GATED = ("encoder.geoembed.mlp.0.weight", "encoder.geoembed.mlp.0.bias", "decoder.geoembed.mlp.0.weight", "decoder.geoembed.mlp.0.bias")


@pytest.mark.parametrize("mode", ["default", "gemm_fp32_tiles", "attention_fp32"])
def test_c2_bench_config_forward_loss_grads_vs_oracle(c2, mode):
    """the bench line's workload, tile heuristics and all: 25 split-bf16 GEMM launches, 256-query attention workgroups,
    split-K weight gradients -- against the oracle at full size.  The model runs its OWN radius search here (as in bench.py).
      * vs the oracle with float64 geometry statistics: output, loss and EVERY gradient tensor within the bar;
      * vs the plain fp32 oracle: output, loss and every gradient tensor within the bar, except the four statistics-gated
        tensors, which must be within twice the oracle's own fp32-vs-fp64 movement."""
    from gaot_amd import ops
    old_g = ops.set_gemm_mode(1) if mode == "gemm_fp32_tiles" else None
    old_a = ops.set_attention_split(0) if mode == "attention_fp32" else None
    try:
        m = _c2_model(c2)
        kw = dict(latent_tokens_coord=c2.lat.to(dev()), xcoord=c2.x.to(dev()))
        check_step(m, (c2.loss64, c2.grads64, c2.pred64), kw, c2.p.to(dev()), c2.t.to(dev()), f"C2 {mode} vs oracle(f64 statistics)", noise=c2.noise64)
        e_out = rel_l2(m(pndata=c2.p.to(dev()), **kw).detach().cpu(), c2.pred)
        errs = grad_errors(m, c2.grads, c2.noise)
        top = max(float(g.double().norm()) for g in c2.grads.values())
        own = {k: float((c2.grads[k].double() - c2.grads64[k].double()).norm()) / float(c2.grads[k].double().norm()) for k in STATS_GATED}
        plain = {k: v for k, v in errs.items() if k not in STATS_GATED}
        worst = max(plain, key=plain.get)
        print(f"[C2 {mode} vs plain fp32 oracle] out rel-L2 {e_out:.2e}  worst grad rel-L2 {plain[worst]:.2e} ({worst}); statistics-gated: "
              + ", ".join(f"{k.split('.')[0][:3]}.{k.split('.')[-1][0]} {errs[k]:.1e} (oracle's own {own[k]:.1e})" for k in STATS_GATED))
        assert e_out < OUT_TOL and plain[worst] < GRAD_TOL, (worst, plain[worst])
        for k in STATS_GATED:
            assert errs[k] < max(GRAD_TOL, 2 * own[k]), (k, errs[k], own[k])
        # the same four tensors against the REFERENCE ITSELF (tests/golden/c2_stats_gates.npz: the imported reference run in float32
        # and in float64 on these weights and this batch): within the 1e-4 bar of its float64 gradients, where the reference's own
        # float32 gradients are 2.4e-4 / 1.5e-4 away.  With this every gradient tensor of the bench configuration faces the same
        # bar against an exported reference vector: the fp32 reference for 68 tensors, the float64 reference for these four.
        from tests._golden import StatsGates
        fx = StatsGates()
        assert fx.same_weights(c2.sd)
        named = dict(m.named_parameters())
        e_ref64 = fx.err({k: named[k].grad for k in STATS_GATED}, fx.g64)
        e_ref32 = fx.err({k: named[k].grad for k in STATS_GATED}, fx.g32)
        print(f"[C2 {mode} statistics-gated vs the reference] " + ", ".join(
            f"{k.split('.')[0][:3]}.{k.split('.')[-1][0]} {e_ref64[k]:.1e} vs ref-f64 / {e_ref32[k]:.1e} vs ref-f32 (reference's own f32-f64 movement {fx.move[k]:.1e})"
            for k in STATS_GATED))
        assert max(e_ref64.values()) < GRAD_TOL, e_ref64
        for cache, want in ((m.encoder.neighbor_cache, c2.enc), (m.decoder.neighbor_cache, c2.dec)):
            nb = list(cache.values())[0][0]                         # the HIP cell list built the oracle's (exact) graph
            assert torch.equal(nb["neighbors_row_splits"].cpu(), want[0][1]) and torch.equal(nb["neighbors_index"].cpu(), want[0][0])
    finally:
        if old_g is not None:
            ops.set_gemm_mode(old_g)
        if old_a is not None:
            ops.set_attention_split(old_a)


def test_c2_default_gradients_stay_within_the_reference_fp32_rounding(c2):
    """What "gradient parity" means for the shipped default, derived from the reference instead of a free 1e-4: measured against the
    oracle in float64 end to end, EVERY gradient tensor of the default path (the fp32-level "f32" precision) is at most 3x as far from
    float64 as the reference's own fp32 arithmetic (the fp32 oracle) is on that tensor, plus 1e-6 (tensors the fp32 oracle happens to
    hit exactly); the output within 2e-6.  Through the TRAINING path (TrainStep: grouped weight gradients, deferred column sums).
    The opt-in bf16x2 precision does NOT meet this bar (4-15x the reference's distance, see test_c2_bf16x2_variant_error_budget)."""
    from gaot_amd import ops
    from gaot_amd.trainer import TrainStep
    assert ops.precision() == "f32"
    m = _c2_model(c2)
    ts = TrainStep(m, lr=8e-4, weight_decay=1e-5, use_graph=False)
    ts.bind(c2.p.to(dev()), c2.t.to(dev()), latent_tokens_coord=c2.lat.to(dev()), xcoord=c2.x.to(dev()))
    with torch.no_grad():
        pred = m(pndata=c2.p.to(dev()), latent_tokens_coord=c2.lat.to(dev()), xcoord=c2.x.to(dev()))
    ts._forward_backward()
    torch.cuda.synchronize()
    # per tensor, relative to ITS OWN norm (no floor): the HIP path's distance from float64 against the fp32 oracle's
    named = dict(m.named_parameters())
    nrm = {k: float(g.norm()) for k, g in c2.gradsd.items()}
    topd = max(nrm.values())
    rel = lambda a, k: float((a.detach().cpu().double() - c2.gradsd[k]).norm()) / max(nrm[k], 1e-9 * topd)
    hip = {k: rel(named[k].grad if named[k].grad is not None else torch.zeros_like(named[k]), k) for k in c2.gradsd}
    own = {k: rel(c2.grads[k], k) for k in c2.gradsd}
    e_out = float((pred.detach().cpu().double() - c2.predd).norm() / c2.predd.norm())
    ratio = {k: hip[k] / (3 * own[k] + 1e-6) for k in hip}
    worst = max(ratio, key=ratio.get)
    wk = max(hip, key=hip.get)
    print(f"[C2 default vs float64 oracle] out {e_out:.2e}; worst gradient tensor {hip[wk]:.2e} ({wk}; fp32 oracle there {own[wk]:.2e}); "
          f"tightest tensor {worst}: {hip[worst]:.2e} against a bar of {3 * own[worst] + 1e-6:.2e}")
    assert e_out < 2e-6, e_out
    assert ratio[worst] <= 1.0, (worst, hip[worst], own[worst])
    ts.bucket.clear()


def test_c2_bf16x2_variant_error_budget(c2, bf16x2):
    """the opt-in two-piece precision at the bench configuration, against the float64 oracle: output within the 1e-5 bar of north_star
    (measured 4.3e-7), every gradient tensor within 2e-5 (measured 6.6e-6) -- and it is NOT at the fp32 level: its worst tensor is more
    than 2e-6 off (the default: < 1e-6).  bench.py reports it as `variants.bf16x2` with these numbers, never as the headline."""
    from gaot_amd import ops
    from gaot_amd.trainer import TrainStep
    assert ops.precision() == "bf16x2"
    m = _c2_model(c2)
    ts = TrainStep(m, lr=8e-4, weight_decay=1e-5, use_graph=False)
    ts.bind(c2.p.to(dev()), c2.t.to(dev()), latent_tokens_coord=c2.lat.to(dev()), xcoord=c2.x.to(dev()))
    with torch.no_grad():
        pred = m(pndata=c2.p.to(dev()), latent_tokens_coord=c2.lat.to(dev()), xcoord=c2.x.to(dev()))
    ts._forward_backward()
    torch.cuda.synchronize()
    hip = grad_errors(m, c2.gradsd)
    wk = max(hip, key=hip.get)
    e_out = float((pred.detach().cpu().double() - c2.predd).norm() / c2.predd.norm())
    print(f"[C2 bf16x2 vs float64 oracle] out {e_out:.2e}; worst gradient tensor {hip[wk]:.2e} ({wk})")
    assert e_out < OUT_TOL and 2e-6 < hip[wk] < 2e-5, (e_out, wk, hip[wk])
    ts.bucket.clear()


def test_c2_radius_graph_backends_differ_only_at_the_boundary(c2):
    """`native` (cdist <= r) vs the exact-difference test at the bench geometry: a handful of the 55.6 k pairs, every one
    within 2e-6 of the radius -- the reference's own backends disagree there, the HIP builder follows the exact ones."""
    import numpy as np
    from oracle import gaot_oracle as O
    for data, q, exact in ((c2.x, c2.lat, c2.enc[0]), (c2.lat, c2.x, c2.dec[0])):
        nat = O.radius_csr(data, q, 0.033)
        pairs = lambda c: set(zip(np.repeat(np.arange(q.shape[0]), np.diff(c[1].numpy())).tolist(), c[0].tolist()))
        diff = pairs(nat) ^ pairs(exact)
        assert len(diff) < 20
        for qi, di in diff:
            assert abs(float((q[qi].double() - data[di].double()).norm()) - 0.033) < 2e-6


@pytest.mark.parametrize("graph", [False, True])
def test_c2_trainstep_post_adamw_weights(c2, graph):
    """TrainStep (flat HIP AdamW; eager and hipGraph replay) at the bench configuration:
    (a) the update applied to the HIP gradients is AdamW's formula to fp32 rounding;
    (b) the step Delta-w agrees with the oracle's per tensor (first step: -lr*(g/(|g|+eps) + wd*w), a sign-like function of g,
        so entries with |g| ~ eps move with rounding noise -- compared in rel-L2 over the tensor against the float64-statistics oracle, 5e-4; measured 1.0e-4)."""
    from gaot_amd.trainer import TrainStep
    from oracle import gaot_oracle as O
    m = _c2_model(c2)
    ts = TrainStep(m, lr=8e-4, weight_decay=1e-5, use_graph=graph)
    ts.bind(c2.p.to(dev()), c2.t.to(dev()), latent_tokens_coord=c2.lat.to(dev()), xcoord=c2.x.to(dev()))
    loss = ts.step()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(c2.loss)) < LOSS_TOL * abs(float(c2.loss))
    worst_a = worst_b = 0.0
    for k, prm in m.named_parameters():
        w0, g = c2.sd[k], prm.grad.detach().cpu()
        want, _, _ = O.adamw_update(w0, g, torch.zeros_like(w0), torch.zeros_like(w0), 1, 8e-4, 1e-5)
        got = prm.detach().cpu()
        worst_a = max(worst_a, float((got - want).abs().max()))
        dw, dw_ref = (got - w0).double(), (c2.new_sd64[k] - w0).double()
        if w0.numel() >= 64:
            worst_b = max(worst_b, float((dw - dw_ref).norm() / dw_ref.norm()))
    print(f"[C2 adamw graph={graph}] max |w - adamw(w0, g_hip)| {worst_a:.2e}; worst rel-L2 of the step vs oracle {worst_b:.2e}")
    assert worst_a < 2e-7 and worst_b < 5e-4


# ------------------------------------------------------------------------------------------------ C3: skewed meshes, vx mode
def _c3_case(B=4, N=8192, spread=0.15, seed=0, **magno_kw):
    from oracle import gaot_oracle as O
    model, sd, ocfg = make_model(3, 1, [64, 64], seed=seed, **magno_kw)
    g = torch.Generator().manual_seed(seed)
    lat = grid([64, 64])
    x = torch.stack([naca_points(N, g, spread) for _ in range(B)])
    p, tgt = torch.randn(B, N, 3, generator=g), torch.randn(B, N, 1, generator=g)
    enc = [[O.radius_csr(x[b], lat, 0.033)] for b in range(B)]
    dec = [[O.radius_csr(lat, x[b], 0.033)] for b in range(B)]
    return model, sd, ocfg, lat, x, p, tgt, enc, dec


def test_c3_naca_vx_degree_skew_vs_oracle():
    """BASELINE configs[2]: per-sample airfoil-like meshes through the block-diagonal batched CSR; rows of > 256 edges sit
    next to empty rows in the encoder CSR, and the same skew appears in the decoder's TRANSPOSED CSR (backward)."""
    from oracle import gaot_oracle as O
    model, sd, ocfg, lat, x, p, tgt, enc, dec = _c3_case()
    deg = torch.cat([e[0][1][1:] - e[0][1][:-1] for e in enc])
    assert int(deg.max()) > 256 and int((deg == 0).sum()) > 1000, (int(deg.max()), int((deg == 0).sum()))
    batch = dict(latent=lat, xcoord=x, pndata=p, target=tgt, encoder_nbrs=enc, decoder_nbrs=dec)
    loss, grads, _, _, pred = O.train_step(sd, ocfg, batch, return_pred=True)
    model.to(dev()).train()
    kw = dict(latent_tokens_coord=lat.to(dev()), xcoord=x.to(dev()),
              encoder_nbrs=[[csr_dict(c) for c in row] for row in enc], decoder_nbrs=[[csr_dict(c) for c in row] for row in dec])
    check_step(model, (loss, grads, pred), kw, p.to(dev()), tgt.to(dev()), f"C3 vx B=4 max degree {int(deg.max())}", noise=fp32_noise(sd, ocfg, batch, grads))
    # a re-shuffled batch (what a shuffling DataLoader hands over next step): permuted samples give permuted outputs
    perm = [2, 0, 3, 1]
    kw2 = dict(latent_tokens_coord=kw["latent_tokens_coord"], xcoord=x[perm].to(dev()),
               encoder_nbrs=[kw["encoder_nbrs"][i] for i in perm], decoder_nbrs=[kw["decoder_nbrs"][i] for i in perm])
    with torch.no_grad():
        y = model(pndata=p.to(dev()), **kw)
        y2 = model(pndata=p[perm].to(dev()), **kw2)
    assert rel_l2(y2.cpu(), y[perm].cpu()) < 1e-6


def test_c3_named_size_batch16_of_8192_nodes_vs_oracle():
    """BASELINE configs[2] AT ITS NAMED SIZE: batch 16 of 8 192-node airfoil-like meshes (vx).  The block-diagonal union has
    131 072 sources, 65 536 latent rows and ~445 k edges per direction: the edge-partitioned kernels' chunking, the composed plans and
    the K slabs of the 16 384-token GEMMs all run at the size the bench reports.  The oracle loops over the samples as the
    reference does (magno.py:356-413).  Also: TrainStep under hipGraph takes the same first step (loss) on this batch."""
    from oracle import gaot_oracle as O
    model, sd, ocfg, lat, x, p, tgt, enc, dec = _c3_case(B=16, N=8192)
    E = sum(int(e[0][0].numel()) for e in enc)
    deg = torch.cat([e[0][1][1:] - e[0][1][:-1] for e in enc])
    assert E > 400_000 and int(deg.max()) > 256 and int((deg == 0).sum()) > 30_000, (E, int(deg.max()), int((deg == 0).sum()))
    batch = dict(latent=lat, xcoord=x, pndata=p, target=tgt, encoder_nbrs=enc, decoder_nbrs=dec)
    loss, grads, _, _, pred = O.train_step(sd, ocfg, batch, return_pred=True)
    model.to(dev()).train()
    kw = dict(latent_tokens_coord=lat.to(dev()), xcoord=x.to(dev()),
              encoder_nbrs=[[csr_dict(c) for c in row] for row in enc], decoder_nbrs=[[csr_dict(c) for c in row] for row in dec])
    check_step(model, (loss, grads, pred), kw, p.to(dev()), tgt.to(dev()), f"C3 vx B=16 x 8192, {E} encoder edges, max degree {int(deg.max())}",
               noise=fp32_noise(sd, ocfg, batch, grads))
    from gaot_amd.trainer import TrainStep
    ts = TrainStep(model, lr=8e-4, weight_decay=1e-5, use_graph=True)
    ts.bind(p.to(dev()), tgt.to(dev()), **kw)
    l0 = float(ts.step())
    assert abs(l0 - float(loss)) < LOSS_TOL * abs(float(loss))


def _drawn_lists(drawn, vx, B, N, M):
    """plan.DROP_RECORD entries of ONE training pass (encoder draw, then decoder draw) as the oracle's neighbour lists.  vx: the draw is made on
    the batch's block-diagonal union (per edge / per row, i.e. per sample); split it back into the per-sample graphs."""
    assert len(drawn) == 2
    if not vx:
        return [drawn[0]], [drawn[1]]

    def split(rec, n_src, q_each):
        idx, sp = rec
        out = []
        for b in range(B):
            lo, hi = int(sp[b * q_each]), int(sp[(b + 1) * q_each])
            out.append([(idx[lo:hi] - b * n_src, sp[b * q_each:(b + 1) * q_each + 1] - lo)])
            assert int(out[-1][0][0].min()) >= 0 and int(out[-1][0][0].max()) < n_src
        return out
    return split(drawn[0], N, M), split(drawn[1], M, N)


@pytest.mark.parametrize("vx", [False, True])
def test_c3_max_neighbors_sampling_on_gpu(vx, monkeypatch):
    """row A12 on the device: training-time neighbour sub-sampling (sampling_strategy='max_neighbors') runs inside the
    model; the CSR lists it drew are recorded and handed to the oracle, which must then agree on output, loss and gradients."""
    from gaot_amd.model.layers import magno as M
    from oracle import gaot_oracle as O
    B, N = 2, 4096
    model, sd, ocfg, lat, x, p, tgt, enc, dec = _c3_case(B=B, N=N, spread=0.2, seed=3, sampling_strategy="max_neighbors", max_neighbors=6)
    if not vx:
        x = x[0]
        enc, dec = [enc[0][0]], [dec[0][0]]
    from gaot_amd import plan as P
    drawn = []
    monkeypatch.setattr(P, "DROP_RECORD", drawn)
    model.to(dev()).train()
    todev = (lambda rows: [[csr_dict(c) for c in row] for row in rows]) if vx else (lambda rows: [csr_dict(c) for c in rows])
    kw = dict(latent_tokens_coord=lat.to(dev()), xcoord=x.to(dev()), encoder_nbrs=todev(enc), decoder_nbrs=todev(dec))
    from gaot_amd import ops
    pred = model(pndata=p.to(dev()), **kw)
    loss = ops.mse_loss(pred, tgt.to(dev()))
    loss.backward()
    enc_d, dec_d = _drawn_lists(drawn, vx, B, N, lat.shape[0])
    if vx:                                      # drawn on the union = per sample (magno.py:372-378)
        full = torch.cat([e[0][1][1:] - e[0][1][:-1] for e in enc])
        kept = torch.cat([e[0][1][1:] - e[0][1][:-1] for e in enc_d])
    else:
        full, kept = enc[0][1][1:] - enc[0][1][:-1], enc_d[0][1][1:] - enc_d[0][1][:-1]
    assert int(full.max()) > 6 and int(kept.max()) == 6 and torch.equal(kept, full.clamp(max=6))
    batch = dict(latent=lat, xcoord=x, pndata=p, target=tgt, encoder_nbrs=enc_d, decoder_nbrs=dec_d)
    loss_ref, grads_ref, _, _, pred_ref = O.train_step(sd, ocfg, batch, return_pred=True)
    assert rel_l2(pred.detach().cpu(), pred_ref) < OUT_TOL
    assert abs(float(loss) - float(loss_ref)) < LOSS_TOL * abs(float(loss_ref))
    errs = grad_errors(model, grads_ref, fp32_noise(sd, ocfg, batch, grads_ref))
    assert max(errs.values()) < GRAD_TOL, max(errs, key=errs.get)
    # TrainStep keeps its hipGraph: the draw happens on the device inside the captured step, a fresh subset on every replay
    # (the plain pass's autograd graph must be gone first: its AccumulateGrad nodes live on the default stream and would be pulled into the capture)
    from gaot_amd.trainer import TrainStep
    monkeypatch.setattr(P, "DROP_RECORD", None)
    del pred, loss
    model.zero_grad(set_to_none=True)
    ts = TrainStep(model, lr=0.0, weight_decay=0.0, use_graph=True)
    assert ts.use_graph is True
    ts.bind(p.to(dev()), tgt.to(dev()), **kw)
    losses = [float(ts.step()) for _ in range(4)]
    assert ts._graphs is not None and len(set(losses)) == 4, losses           # same weights (lr = 0), same batch: only the draw differs


@pytest.mark.parametrize("vx", [False, True])
def test_c3_ratio_sampling_on_gpu(vx, monkeypatch):
    """row A12, the other strategy (edge_drop.py:54-68 of the reference): sampling_strategy='ratio' keeps each edge with probability
    sample_ratio, drawn on the device inside the model.  The drawn CSR lists are recorded and handed to the oracle, which must agree on
    output, loss and gradients; the draw itself must be a sub-sequence of every row (order kept), keep about the stated share, differ
    between two forward passes, and be skipped in eval mode."""
    from gaot_amd.model.layers import magno as M
    from oracle import gaot_oracle as O
    B, N, ratio = 2, 4096, 0.6
    model, sd, ocfg, lat, x, p, tgt, enc, dec = _c3_case(B=B, N=N, spread=0.2, seed=4, sampling_strategy="ratio", sample_ratio=ratio)
    if not vx:
        x = x[0]
        enc, dec = [enc[0][0]], [dec[0][0]]
    from gaot_amd import plan as P
    drawn = []
    monkeypatch.setattr(P, "DROP_RECORD", drawn)
    model.to(dev()).train()
    todev = (lambda rows: [[csr_dict(c) for c in row] for row in rows]) if vx else (lambda rows: [csr_dict(c) for c in rows])
    kw = dict(latent_tokens_coord=lat.to(dev()), xcoord=x.to(dev()), encoder_nbrs=todev(enc), decoder_nbrs=todev(dec))
    from gaot_amd import ops
    pred = model(pndata=p.to(dev()), **kw)
    loss = ops.mse_loss(pred, tgt.to(dev()))
    loss.backward()
    first = list(drawn)
    enc_d, dec_d = _drawn_lists(first, vx, B, N, lat.shape[0])
    if vx:
        full_graphs = [e[0] for e in enc] + [d[0] for d in dec]
        drawn_graphs = [e[0] for e in enc_d] + [d[0] for d in dec_d]
    else:
        full_graphs = [enc[0], dec[0]]
        drawn_graphs = [enc_d[0], dec_d[0]]
    for (fi, fs), (ki, ks) in zip(full_graphs, drawn_graphs):
        E, Ek = int(fi.numel()), int(ki.numel())
        assert abs(Ek / E - ratio) < 0.03, (Ek, E)                        # ~55 k draws per graph: 3 sigma is 0.006
        assert int(ks[-1]) == Ek and ks.numel() == fs.numel() and bool(((ks[1:] - ks[:-1]) <= (fs[1:] - fs[:-1])).all())
        for q in (0, 17, int(fs.numel()) // 2, int(fs.numel()) - 2):     # kept neighbours: a sub-sequence of the row, order preserved
            row, sub = fi[fs[q]:fs[q + 1]].tolist(), ki[ks[q]:ks[q + 1]].tolist()
            it = iter(row)
            assert all(v in it for v in sub), (q, row, sub)
    batch = dict(latent=lat, xcoord=x, pndata=p, target=tgt, encoder_nbrs=enc_d, decoder_nbrs=dec_d)
    loss_ref, grads_ref, _, _, pred_ref = O.train_step(sd, ocfg, batch, return_pred=True)
    assert rel_l2(pred.detach().cpu(), pred_ref) < OUT_TOL
    assert abs(float(loss) - float(loss_ref)) < LOSS_TOL * abs(float(loss_ref))
    errs = grad_errors(model, grads_ref, fp32_noise(sd, ocfg, batch, grads_ref))
    assert max(errs.values()) < GRAD_TOL, max(errs, key=errs.get)
    # a second pass draws another subset; eval mode draws none (the full graph: edge_drop.py `if not training`)
    model(pndata=p.to(dev()), **kw)
    assert len(drawn) == 4 and not torch.equal(drawn[0][1], drawn[2][1])
    model.eval()
    with torch.no_grad():
        y_eval = model(pndata=p.to(dev()), **kw)
    assert len(drawn) == 4
    pred_full = O.gaot_forward(sd, ocfg, lat, x, p, encoder_nbrs=enc, decoder_nbrs=dec)
    assert rel_l2(y_eval.cpu(), pred_full) < OUT_TOL
    # TrainStep keeps its hipGraph: a fresh subset on every replay (same weights, same batch: only the draw moves the loss), about the stated share
    from gaot_amd.trainer import TrainStep
    monkeypatch.setattr(P, "DROP_RECORD", None)
    model.train()
    del pred, loss          # (the plain pass's autograd graph must be gone: its AccumulateGrad nodes live on the default stream)
    model.zero_grad(set_to_none=True)
    ts = TrainStep(model, lr=0.0, weight_decay=0.0, use_graph=True)
    assert ts.use_graph is True
    ts.bind(p.to(dev()), tgt.to(dev()), **kw)
    losses = [float(ts.step()) for _ in range(4)]
    assert ts._graphs is not None and len(set(losses)) == 4, losses
    side = model.encoder
    base = (side._static_unions[next(iter(side._static_unions))].plan if vx else P.plan_for(kw["encoder_nbrs"][0], x.shape[0]))
    dp = next(iter(base._drops.values()))
    e_full = sum(int(f[0].numel()) for f in full_graphs[:len(full_graphs) // 2])
    assert abs(int(dp.e_dev.item()) / e_full - ratio) < 0.03


@pytest.mark.parametrize("stress", ["token_1e4", "token_1e6", "channel_1e4", "channel_1e6", "weight_row_1e-6", "all"])
def test_c2_processor_dynamic_range_stress(stress):
    """Model-level dynamic range of the default (fp16-piece) products -- the pattern trained transformers show ("massive activations":
    one token / one channel 1e3-1e6 times larger than the rest on the un-normed operands) and every parity test with N(0, 1) fields
    misses.  The C2 processor (patchify -> patch_linear + positions -> 3 blocks -> unpatchify; 8 x 1 024 tokens of 256) is fed a
    latent field with ONE token (the four latent nodes of one patch of one sample) and / or ONE channel scaled by 1e4 / 1e6, and / or
    carries weight rows scaled by 1e-6 (patch_linear, w1, q_proj, o_proj); loss = MSE against a random target.  Against the oracle in
    float64 end to end, the HIP path's output and EVERY gradient (all processor weights AND the input field) must be within 3x the
    fp32 oracle's own distance + 1e-6 -- the bar of test_c2_default_gradients_stay_within_the_reference_fp32_rounding -- which it meets
    because tiles whose operand rows span more than 2^13 take the per-row second pass (the counter must show it did)."""
    from gaot_amd import ops, _lib
    from oracle import gaot_oracle as O
    lib = _lib.load()
    assert ops.precision() == "f32"
    model, sd, ocfg = make_model(1, 1, [64, 64], seed=11)
    g = torch.Generator().manual_seed(11)
    B = 8                 # (8 192 tokens, as C2: at 4 096 the N = 256 products run on the fp32-MFMA tiles, which need no scales)
    rn = torch.randn(B, 4096, 64, generator=g)
    tgt = torch.randn(B, 4096, 64, generator=g)
    kinds = stress.split("_")
    scale = float(stress.rsplit("_", 1)[1]) if stress != "all" else 1e6
    if kinds[0] in ("token", "all"):
        rn[1, [130, 131, 194, 195]] *= scale               # patch (1, 1) of sample 1: latent nodes (2..3, 2..3) of the 64 x 64 grid
    if kinds[0] in ("channel", "all"):
        rn[:, :, 17] *= (scale if stress != "all" else 1e4)
    if kinds[0] in ("weight", "all"):
        for k, rows in (("patch_linear.weight", [7]), ("processor.encoder_layers.0.ffn.w1.weight", [5, 600]),
                        ("processor.middle_layer.attn.q_proj.weight", [33]), ("processor.decoder_layers.0.attn.o_proj.weight", [2])):
            sd[k] = sd[k].clone()
            sd[k][rows] *= 1e-6
    names = [k for k in sd if k.startswith("processor.") or k.startswith("patch_linear.")]

    def oracle(dtype):
        ps = {k: sd[k].to(dtype).clone().requires_grad_(True) for k in names}
        x = rn.to(dtype).clone().requires_grad_(True)
        out = O.process(ps, ocfg, x)
        loss = torch.mean((out - tgt.to(dtype)) ** 2)
        gs = torch.autograd.grad(loss, [x] + [ps[k] for k in names])
        return out.detach(), {"input": gs[0], **{k: v for k, v in zip(names, gs[1:])}}

    out64, g64 = oracle(torch.float64)
    out32, g32 = oracle(torch.float32)
    model.load_state_dict(sd)
    model.to(dev()).train()
    x = rn.to(dev()).requires_grad_(True)
    lib.gaot_debug_split_redo_count(1)
    # the TRAINING path's launches: gradient slots + every weight gradient from the grouped launch at the end of the pass
    from gaot_amd.trainer import FlatGradBucket
    groups = [gr for m_ in model.modules() if hasattr(m_, "fused_weight_groups") for gr in m_.fused_weight_groups()]
    bucket = FlatGradBucket(list(model.parameters()), groups)
    bucket.clear()
    ops.begin_pass()
    ops.refresh_weight_amax(list(model.parameters()), groups)
    out = model.process(rndata=x)
    loss = ops.mse_loss(out, tgt.to(dev()))
    with ops.deferred_wgrad():
        loss.backward()
    bucket.pack()
    torch.cuda.synchronize()
    redone = int(lib.gaot_debug_split_redo_count(1))
    got = {"input": x.grad, **{k: q.grad for k, q in model.named_parameters() if k in names}}
    top = max(float(v.norm()) for v in g64.values())
    err = lambda a, ref: float((a.detach().cpu().double() - ref).norm()) / max(float(ref.norm()), 1e-3 * top)
    hip = {k: err(got[k], g64[k]) for k in g64}
    own = {k: err(g32[k], g64[k]) for k in g64}
    e_out, o_out = rel_l2(out.detach().cpu(), out64), rel_l2(out32, out64)
    tight = max(hip, key=lambda k: hip[k] / (3 * own[k] + 1e-6))
    print(f"[C2 processor, {stress}] tiles through the per-row pass: {redone}; out {e_out:.2e} (fp32 oracle {o_out:.2e}); tightest gradient {tight}: "
          f"{hip[tight]:.2e} against 3 x {own[tight]:.2e} + 1e-6; worst {max(hip.values()):.2e}")
    assert redone > 0 or not (stress.endswith("1e6") or stress == "all"), redone        # (1e4 = 2^13.3: the edge of the tensor-wide scale's range)
    bucket.clear()
    ops.register_grad_slots([], [])
    assert e_out <= 3 * o_out + 1e-6, (e_out, o_out)
    assert hip[tight] <= 3 * own[tight] + 1e-6, (tight, hip[tight], own[tight])


# ------------------------------------------------------------------------------------------------ C1 and C5
def test_c1_poisson_1k_nodes_batch4_many_empty_tokens():
    """BASELINE configs[0] at its stated shape: 1 024 nodes, batch 4, example model; ~43 % of the 4 096 latent tokens have
    no physical node within the radius (empty CSR rows -> zeros through AGNO, statistics and the geometry embedding)."""
    from oracle import gaot_oracle as O
    model, sd, ocfg = make_model(1, 1, [64, 64], seed=5)
    g = torch.Generator().manual_seed(5)
    lat, x = grid([64, 64]), uniform_points(1024, 2, g)
    p, tgt = torch.randn(4, 1024, 1, generator=g), torch.randn(4, 1024, 1, generator=g)
    enc, dec = [O.radius_csr(x, lat, 0.033)], [O.radius_csr(lat, x, 0.033)]
    deg = enc[0][1][1:] - enc[0][1][:-1]
    assert 0.35 < float((deg == 0).float().mean()) < 0.5
    batch = dict(latent=lat, xcoord=x, pndata=p, target=tgt, encoder_nbrs=enc, decoder_nbrs=dec)
    loss, grads, _, _, pred = O.train_step(sd, ocfg, batch, return_pred=True)
    model.to(dev()).train()
    kw = dict(latent_tokens_coord=lat.to(dev()), xcoord=x.to(dev()), encoder_nbrs=[csr_dict(enc[0])], decoder_nbrs=[csr_dict(dec[0])])
    check_step(model, (loss, grads, pred), kw, p.to(dev()), tgt.to(dev()), "C1 1k nodes B=4", noise=fp32_noise(sd, ocfg, batch, grads))


@pytest.mark.parametrize("n_points", [16384, 65536])
def test_c5_3d_cloud_4096_tokens_headdim48(n_points):
    """BASELINE configs[4]: 3-D surface cloud, 32^3 latent grid -> 4 096 tokens of width 8*48 = 384, 8 heads x 48 (the head_dim-64
    split-bf16 attention kernels), 48 lifting channels (kernel MLP and geometry-embedding chain with layers narrower than 64),
    92 % empty latent rows.  65 536 points = the NAMED size (308 k edges per direction, encoder rows of > 600 edges); 16 384 points
    = the same shape at a quarter of the cloud."""
    from oracle import gaot_oracle as O
    model, sd, ocfg = make_model(3, 1, [32, 32, 32], d=3, C=48, hidden=384, heads=8, radius=0.067, seed=7)
    g = torch.Generator().manual_seed(7)
    lat, x = grid([32, 32, 32]), shell_points(n_points, g)
    p, tgt = torch.randn(1, n_points, 3, generator=g), torch.randn(1, n_points, 1, generator=g)
    enc, dec = [O.radius_csr(x, lat, 0.067)], [O.radius_csr(lat, x, 0.067)]
    batch = dict(latent=lat, xcoord=x, pndata=p, target=tgt, encoder_nbrs=enc, decoder_nbrs=dec)
    loss, grads, _, _, pred = O.train_step(sd, ocfg, batch, return_pred=True)
    # the same algorithm evaluated in float64 throughout: the instrument for the tensors right behind the geometry statistics'
    # ReLU gates.  On this geometry the fp32 oracle's OWN gradient of decoder.geoembed.mlp.0.weight is 3e-4 (relative L2) away
    # from its float64 evaluation (gates within rounding of zero flip); the fused row-wise MLP kernel lands 9e-8 from the
    # float64 value, the GEMM-chain path lands on the fp32 oracle's.  A tensor passes when it is within the bar of EITHER.
    sd64 = {k: v.double() for k, v in sd.items()}
    b64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
    _, grads64, _, _, _ = O.train_step(sd64, ocfg, b64, return_pred=True)
    model.to(dev()).train()
    kw = dict(latent_tokens_coord=lat.to(dev()), xcoord=x.to(dev()), encoder_nbrs=[csr_dict(enc[0])], decoder_nbrs=[csr_dict(dec[0])])
    from gaot_amd import ops
    model.zero_grad(set_to_none=True)
    out = model(pndata=p.to(dev()), **kw)
    l = ops.mse_loss(out, tgt.to(dev()))
    l.backward()
    torch.cuda.synchronize()
    assert rel_l2(out.detach().cpu(), pred) < OUT_TOL
    assert abs(float(l.detach()) - float(loss)) < LOSS_TOL * abs(float(loss))
    noise = {k: float((grads[k].double() - grads64[k]).norm()) for k in grads64}
    e32, e64 = grad_errors(model, grads, noise), grad_errors(model, grads64, noise)
    worst = max(e32, key=lambda k: min(e32[k], e64[k]))
    print(f"[C5 3-D {n_points} points, {int(enc[0][0].numel())} edges, 4096 tokens head_dim 48] worst gradient tensor {worst}: {e32[worst]:.2e} vs the fp32 oracle, {e64[worst]:.2e} vs its float64 evaluation; "
          + ", ".join(f"{k.split('.')[0][:3]}.{k.split('.')[-1][0]} {e32[k]:.1e}/{e64[k]:.1e}" for k in STATS_GATED))
    for k in e32:
        assert min(e32[k], e64[k]) < GRAD_TOL, (k, e32[k], e64[k])


# ------------------------------------------------------------------------------------------------ caches vs in-place optimizers
def test_train_eval_train_eval_sees_new_weights():
    """The reference's loop (eval_every_eps): validate / roll out, train some more, validate again on the SAME neighbour
    dicts and coordinates.  FlatAdamW (and its hipGraph replay) update weights through raw pointers, which does not move
    Parameter._version: the no_grad caches (AGNO kernel values, geoembed row bias, the rollout hipGraph) must still notice."""
    from gaot_amd.model.gaot import GAOT
    from gaot_amd.trainer import TrainStep
    model, sd, _ = make_model(3, 2, [32, 32], C=32, hidden=128, heads=4, radius=0.08, precompute=False, seed=11)
    model.to(dev())
    g = torch.Generator().manual_seed(11)
    lat, x = grid([32, 32]).to(dev()), uniform_points(1500, 2, g).to(dev())
    p, tgt = torch.randn(2, 1500, 3, generator=g).to(dev()), torch.randn(2, 1500, 2, generator=g).to(dev())
    def evaluate(m):
        m.eval()
        with torch.no_grad():
            y = m(latent_tokens_coord=lat, xcoord=x, pndata=p)
        m.train()
        return y.clone()

    for graph in (False, True):
        m = GAOT(3, 2, model_cfg(model))
        m.load_state_dict(sd)
        m.to(dev()).train()
        ts = TrainStep(m, lr=5e-3, weight_decay=1e-5, use_graph=graph)
        ts.bind(p, tgt, latent_tokens_coord=lat, xcoord=x)
        y0 = evaluate(m)
        for _ in range(3):
            ts.step()
        y1 = evaluate(m)
        fresh = GAOT(3, 2, model_cfg(model))
        fresh.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
        fresh.to(dev())
        y1_ref = evaluate(fresh)
        assert rel_l2(y1.cpu(), y0.cpu()) > 1e-3                    # training moved the prediction ...
        assert rel_l2(y1.cpu(), y1_ref.cpu()) < 1e-6, graph         # ... and the evaluation reflects the CURRENT weights


def model_cfg(model):
    from gaot_amd.model.layers.attn import TransformerConfig, AttentionConfig
    enc = model.encoder.config
    blk = model.processor.blocks_in_order()[0]
    hidden = blk.attn.q_proj.weight.shape[0]
    tcfg = TransformerConfig(patch_size=model.patch_size, hidden_size=hidden,
                             attn_config=AttentionConfig(num_heads=blk.attn.num_heads, num_kv_heads=blk.attn.num_kv_heads))
    return NS(args=NS(magno=enc, transformer=tcfg), latent_tokens_size=[model.H, model.W] + ([model.D] if model.D else []))


def test_rollout_graph_follows_weight_updates():
    """autoregressive_predict's captured step must be re-captured (or refreshed) after raw-pointer weight updates"""
    import numpy as np
    from gaot_amd.model.gaot import GAOT
    from gaot_amd.trainer import TrainStep
    model, sd, _ = make_model(3, 1, [32, 32], C=32, hidden=128, heads=4, radius=0.08, precompute=False, seed=12)
    g = torch.Generator().manual_seed(12)
    lat, x = grid([32, 32]).to(dev()), uniform_points(1200, 2, g).to(dev())
    xb = torch.randn(2, 1200, 1, generator=g).to(dev())                   # u only; the model input is [u, t0, dt] = 3 columns
    p3, tgt = torch.randn(2, 1200, 3, generator=g).to(dev()), torch.randn(2, 1200, 1, generator=g).to(dev())
    stats = {"u": {"mean": torch.zeros(1), "std": torch.ones(1)}, "start_time": {"mean": 0.0, "std": 1.0}, "time_diffs": {"mean": 0.0, "std": 1.0}}
    tv, ti = np.linspace(0, 1, 5), np.arange(4)
    roll = lambda m: m.autoregressive_predict(x_batch=xb, time_indices=ti, t_values=tv, stats=stats, stepper_mode="output",
                                              latent_tokens_coord=lat, fixed_coord=x)
    m = model.to(dev())
    m.eval()
    r0 = roll(m)
    m.train()
    ts = TrainStep(m, lr=5e-3, weight_decay=1e-5, use_graph=True)
    ts.bind(p3, tgt, latent_tokens_coord=lat, xcoord=x)
    for _ in range(3):
        ts.step()
    m.eval()
    r1 = roll(m)
    fresh = GAOT(3, 1, model_cfg(m))
    fresh.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
    fresh.to(dev()).eval()
    r1_ref = roll(fresh)
    assert rel_l2(r1.cpu(), r0.cpu()) > 1e-3 and rel_l2(r1.cpu(), r1_ref.cpu()) < 1e-6


def test_lr_schedule_reaches_the_captured_optimizer():
    """torch LR schedulers attach to FlatAdamW (a torch.optim.Optimizer with one param_group) and the captured update reads
    lr from device memory: three graph-replayed steps under StepLR equal three torch.optim.AdamW steps under the same schedule."""
    from gaot_amd.model.gaot import GAOT
    from gaot_amd.trainer import TrainStep
    model, sd, _ = make_model(1, 1, [16, 16], C=16, hidden=64, heads=2, radius=0.2, precompute=False, seed=13)
    g = torch.Generator().manual_seed(13)
    lat, x = grid([16, 16]).to(dev()), uniform_points(300, 2, g).to(dev())
    p, tgt = torch.randn(2, 300, 1, generator=g).to(dev()), torch.randn(2, 300, 1, generator=g).to(dev())
    ms = []
    for _ in range(2):
        m = GAOT(1, 1, model_cfg(model))
        m.load_state_dict(sd)
        ms.append(m.to(dev()).train())
    ts = TrainStep(ms[0], lr=4e-3, weight_decay=1e-2, use_graph=True)
    ts.bind(p, tgt, latent_tokens_coord=lat, xcoord=x)
    sched_a = torch.optim.lr_scheduler.StepLR(ts.opt, step_size=1, gamma=0.5)
    opt_b = torch.optim.AdamW(ms[1].parameters(), lr=4e-3, weight_decay=1e-2)
    sched_b = torch.optim.lr_scheduler.StepLR(opt_b, step_size=1, gamma=0.5)
    for _ in range(3):
        ts.step()
        sched_a.step()
        opt_b.zero_grad()
        torch.nn.functional.mse_loss(ms[1](latent_tokens_coord=lat, xcoord=x, pndata=p), tgt).backward()
        opt_b.step()
        sched_b.step()
    assert ts.opt.param_groups[0]["lr"] == opt_b.param_groups[0]["lr"] == 5e-4
    for (k, a), (_, b) in zip(ms[0].named_parameters(), ms[1].named_parameters()):
        assert float((a.detach() - b.detach()).abs().max()) < 2e-5, k
    # and a frozen-lr twin would NOT match: the schedule really reached the kernel
    m3 = GAOT(1, 1, model_cfg(model))
    m3.load_state_dict(sd)
    ts3 = TrainStep(m3.to(dev()).train(), lr=4e-3, weight_decay=1e-2, use_graph=True)
    ts3.bind(p, tgt, latent_tokens_coord=lat, xcoord=x)
    for _ in range(3):
        ts3.step()
    assert max(float((a.detach() - b.detach()).abs().max()) for a, b in zip(ms[0].parameters(), m3.parameters())) > 1e-3


def test_geometry_caches_hit_by_content_not_identity():
    """the reference trainer re-uploads the unchanged coordinates every step (static_trainer.py:167-170): new tensor objects,
    old bytes.  The plan's coordinate-derived arrays must be reused (same storage, no re-allocation) with identical results;
    and when the bytes DO change under the same shape, the arrays are refreshed in place (device-side guard, no host sync)."""
    from gaot_amd.plan import plan_for
    from oracle import gaot_oracle as O
    model, sd, _ = make_model(1, 1, [32, 32], C=32, hidden=128, heads=4, radius=0.08, precompute=False, seed=21)
    model.to(dev()).train()
    g = torch.Generator().manual_seed(21)
    lat, x = grid([32, 32]), uniform_points(2000, 2, g)
    p = torch.randn(2, 2000, 1, generator=g).to(dev())
    y1 = model(latent_tokens_coord=lat.to(dev()), xcoord=x.to(dev()), pndata=p)
    nb = list(model.encoder.neighbor_cache.values())[0][0]
    # the forward runs over the module's list renumbered to the patch-major latent order (GAOT._patch_major): that list's plan holds the arrays
    assert "_gaot_amd_renumbered" in nb
    nb = nb["_gaot_amd_renumbered"][1]
    lat = lat[model._latent_order(dev())[0].cpu()]
    plan = plan_for(nb, 2000)
    ptrs = {k: v["val"].data_ptr() for k, v in plan._coord_cache.items()}
    assert set(ptrs) == {"feat", "cos", "stats1"}
    y2 = model(latent_tokens_coord=grid([32, 32]).to(dev()), xcoord=x.to(dev()), pndata=p)           # fresh uploads, same bytes
    assert torch.equal(y1, y2)
    assert {k: v["val"].data_ptr() for k, v in plan._coord_cache.items()} == ptrs and plan_for(nb, 2000) is plan
    # same shapes, different bytes: the guarded kernels recompute in place
    x2 = (x + 0.001 * torch.randn(x.shape, generator=g)).clamp(-1, 1)
    xs, ls = x2.to(dev()), lat.to(dev())
    feat = plan.edge_features(xs, ls)
    cos = plan.cosine_attention(xs, ls)
    st = plan.geo_stats(xs, ls)
    assert feat.data_ptr() == ptrs["feat"] and st.data_ptr() == ptrs["stats1"]
    idx, sp = nb["neighbors_index"].cpu(), nb["neighbors_row_splits"].cpu()
    qid, _ = O.edge_query_ids(sp)
    assert torch.equal(feat.cpu(), torch.cat([x2[idx], lat[qid]], dim=1))
    assert rel_l2(st.cpu(), O.geo_stats(x2, lat, (idx, sp))) < 1e-4
    sc = (torch.nn.functional.normalize(lat[qid], dim=-1) * torch.nn.functional.normalize(x2[idx], dim=-1)).sum(-1)
    assert rel_l2(cos.cpu()[:idx.numel()], O.segment_softmax(sc, qid, lat.shape[0])) < 1e-5
    # eval-mode kernel values follow the refreshed geometry (they are cached on the host under no_grad)
    model.eval()
    with torch.no_grad():
        lat0 = grid([32, 32]).to(dev())
        e_old = model.encode(x.to(dev()), p, lat0, None)
        e_new = model.encode(x2.to(dev()), p, lat0, None)
        e_new2 = model.encode(x2.to(dev()), p, lat0, None)
    assert rel_l2(e_new.cpu(), e_old.cpu()) > 1e-4 and torch.equal(e_new, e_new2)


def test_vx_union_composed_from_per_sample_plans_equals_union_planned_afresh():
    """vx mode re-planning (reference magno.py:356-413 loops over samples): the block-diagonal union of a batch is either planned
    over the concatenated CSR (fresh dicts) or COMPOSED from per-sample plans by concatenation with offsets (dicts seen before,
    any batch order).  Both must give the same int32 CSR, transposed CSR (edge ids ascending per source -- also for the
    350-edge rows of these skewed meshes), per-sample standardised statistics, edge features and attention weights."""
    from gaot_amd.plan import MergedGeometry, merged_geometry
    from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch
    g = torch.Generator().manual_seed(31)
    B, N = 5, 4096
    lat = grid([64, 64]).to(dev())
    xs = [naca_points(N, g, 0.12).to(dev()) for _ in range(B)]
    ns = NeighborSearch("native")
    for src_of, dst_of in ((lambda b: xs[b], lambda b: lat), (lambda b: lat, lambda b: xs[b])):      # encoder- and decoder-shaped
        dicts = [ns(src_of(b), dst_of(b), 0.033) for b in range(B)]
        order = [3, 0, 4, 1, 2]
        fresh = MergedGeometry([dicts[i] for i in order], [src_of(i) for i in order], [dst_of(i) for i in order])
        comp = MergedGeometry([dicts[i] for i in order], [src_of(i) for i in order], [dst_of(i) for i in order], build_parts=True)
        assert not fresh.composed and comp.composed
        a, b = fresh.plan, comp.plan
        assert (a.Q, a.E, a.n_src) == (b.Q, b.E, b.n_src)
        for name in ("index", "splits", "edge_query", "t_splits", "t_edge"):
            assert torch.equal(getattr(a, name)[:a.E if name in ("index", "edge_query", "t_edge") else None], getattr(b, name)[:a.E if name in ("index", "edge_query", "t_edge") else None]), name
        assert torch.equal(a.deg, b.deg)
        te, ts_ = b.t_edge[:b.E].cpu(), b.t_splits.cpu()
        long_rows = 0
        for j in torch.nonzero((ts_[1:] - ts_[:-1]) > 64).flatten().tolist()[:50]:      # the long rows went through the rank sort
            seg = te[ts_[j]:ts_[j + 1]]
            assert bool((seg[1:] > seg[:-1]).all())
            long_rows += 1
        assert torch.equal(fresh.geo_stats(), comp.geo_stats())
        assert torch.equal(a.edge_features(fresh.src, fresh.dst), b.edge_features(comp.src, comp.dst))
        assert torch.equal(a.cosine_attention(fresh.src, fresh.dst), b.cosine_attention(comp.src, comp.dst))
        # per-sample statistics inside the union == statistics of each sample alone
        from gaot_amd.plan import plan_for
        alone = torch.cat([plan_for(dicts[i], src_of(i).shape[0]).geo_stats(src_of(i), dst_of(i)) for i in order])
        assert rel_l2(comp.geo_stats().cpu(), alone.cpu()) < 1e-6
    # the cache: first sight of a batch plans the union afresh, a later batch of the same dicts (other order) composes
    dicts = [ns(xs[b], lat, 0.033) for b in range(B)]
    x_all = torch.stack(xs)
    m1 = merged_geometry(dicts, xs, [lat] * B, parents=(x_all, lat))
    perm = [4, 2, 0, 1, 3]
    m2 = merged_geometry([dicts[i] for i in perm], [xs[i] for i in perm], [lat] * B, parents=(x_all[perm], lat))
    assert not m1.composed and m2.composed


def test_vx_shuffling_loader_does_not_pile_up_unions():
    """Under a shuffling loader (the reference's default, data_utils.py:272-294) every step brings a batch composition that never comes
    back.  The union cache keeps only a few most recently used unions and releases what it evicts (the derived arrays sit in reference
    cycles: without the release they wait for Python's cyclic collector), so device memory stays flat from step to step -- no
    collector run needed."""
    import gc, itertools
    from gaot_amd import plan as P
    from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch
    g = torch.Generator().manual_seed(5)
    B, N = 4, 2048
    lat = grid([32, 32]).to(dev())
    xs = [naca_points(N, g, 0.12).to(dev()) for _ in range(B)]
    ns = NeighborSearch("native")
    dicts = [ns(xs[b], lat, 0.066) for b in range(B)]
    x_all = torch.stack(xs)
    perms = list(itertools.permutations(range(B)))

    def step(perm):
        xp = x_all[list(perm)]
        mg = P.merged_geometry([dicts[i] for i in perm], [xp[i] for i in range(B)], [lat] * B, parents=(xp, lat))
        mg.geo_stats(); mg.plan.edge_features(mg.src, mg.dst); mg.plan.cosine_attention(mg.src, mg.dst)
        return mg
    gc.collect()
    gc.disable()
    try:
        for perm in perms[:P._MERGE_CACHE_MAX + 2]:
            step(perm)
        torch.cuda.synchronize()
        m0 = torch.cuda.memory_allocated()
        for perm in perms[P._MERGE_CACHE_MAX + 2:24]:
            step(perm)
        torch.cuda.synchronize()
        grown = torch.cuda.memory_allocated() - m0
    finally:
        gc.enable()
    assert len(P._MERGE_CACHE) <= P._MERGE_CACHE_MAX
    assert grown < (1 << 20), grown          # flat (the old cache of 64 grew by one union per step)


def test_auto_graph_reference_loop_equals_eager_loop():
    """autograph.py: the reference trainer's loop, unchanged (per-step uploads of batch and coordinates, zero_grad, eager call,
    nn.MSELoss, backward, torch.optim.AdamW), runs forward and backward as hipGraph replays from its third step on.  It must give
    what the plain eager HIP path gives: same losses, same weights -- also when the coordinates' CONTENT changes under the same
    shape mid-run (the captured kernels read geometry arrays that the device-side content guard refreshes in place), across an
    evaluation in between, and with gradient accumulation (no zero_grad)."""
    from gaot_amd.model.gaot import GAOT
    model, sd, _ = make_model(2, 1, [32, 32], C=32, hidden=128, heads=4, radius=0.08, precompute=False, seed=41)
    g = torch.Generator().manual_seed(41)
    lat = grid([32, 32])
    x1 = uniform_points(1800, 2, g)
    x2 = (x1 + 0.002 * torch.randn(x1.shape, generator=g)).clamp(-1, 1)          # same shape, other bytes (steps 5..)
    data = [(torch.randn(3, 1800, 2, generator=g), torch.randn(3, 1800, 1, generator=g)) for _ in range(8)]
    runs = {}
    for auto in (True, False):
        m = GAOT(2, 1, model_cfg(model))
        m.load_state_dict(sd)
        m.to(dev()).train()
        m.auto_graph = auto
        opt = torch.optim.AdamW(m.parameters(), lr=2e-3, weight_decay=1e-4)
        lossf = torch.nn.MSELoss()
        losses, evals = [], []
        for i, (p, t) in enumerate(data):
            xc = x1 if i < 5 else x2
            xb, yb, latd, coord = p.to(dev()), t.to(dev()), lat.to(dev()), xc.to(dev())       # fresh device tensors every step
            opt.zero_grad()
            loss = lossf(m(latent_tokens_coord=latd, xcoord=coord, pndata=xb), yb)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
            if i == 3:
                m.eval()
                with torch.no_grad():
                    evals.append(m(latent_tokens_coord=latd, xcoord=coord, pndata=xb).cpu())
                m.train()
        if auto:
            assert len(m._auto_graph_cache) == 1                                             # it did capture
        # another batch size = another captured entry sharing the static coordinate buffers, first seen with the OLD coordinates
        for xc in (x1, x1, x1, x2):
            opt.zero_grad()
            loss = lossf(m(latent_tokens_coord=lat.to(dev()), xcoord=xc.to(dev()), pndata=data[0][0][:2].to(dev())), data[0][1][:2].to(dev()))
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        # gradient accumulation: two backward passes without zero_grad
        opt.zero_grad()
        for p, t in data[:2]:
            lossf(m(latent_tokens_coord=lat.to(dev()), xcoord=x2.to(dev()), pndata=p.to(dev())), t.to(dev())).backward()
        acc = torch.cat([q.grad.detach().reshape(-1) for q in m.parameters()]).cpu()
        runs[auto] = (losses, torch.cat([q.detach().reshape(-1) for q in m.parameters()]).cpu(), evals[0], acc)
    la, wa, ea, ga = runs[True]
    lb, wb, eb, gb = runs[False]
    assert max(abs(a - b) / abs(b) for a, b in zip(la, lb)) < 1e-5, (la, lb)
    assert float((wa - wb).abs().max()) < 2e-5
    assert rel_l2(ea, eb) < 1e-5
    assert rel_l2(ga, gb) < 1e-4


@pytest.mark.parametrize("weighted", [False, True])
def test_multiscale_shared_weights_with_deferred_gradients(weighted):
    """scales = [1, 2]: the reference shares the agno / geoembed / lifting / recovery / projection weights across scales
    (magno.py:277-300), so every such parameter is used TWICE in a forward pass.  Inside a deferral scope (TrainStep eager and graph,
    autograph from its second step) only a parameter used exactly once may have its gradient slice written by the deferred, grouped
    launches (ops._SHARED_SLOTS): gradients of TrainStep and of the auto-graphed plain loop must equal the plain eager backward, and
    the multiscale forward itself is pinned by the golden cases ms_mean / ms_weighted."""
    from gaot_amd import ops
    from gaot_amd.model.gaot import GAOT
    from gaot_amd.trainer import TrainStep
    model, sd, _ = make_model(2, 1, [32, 32], C=32, hidden=128, heads=4, radius=0.08, precompute=False, seed=43, scales=[1.0, 2.0],
                              use_scale_weights=weighted)
    g = torch.Generator().manual_seed(43)
    lat, x = grid([32, 32]), uniform_points(2400, 2, g)
    p, t = torch.randn(4, 2400, 2, generator=g), torch.randn(4, 2400, 1, generator=g)
    kw = dict(latent_tokens_coord=lat.to(dev()), xcoord=x.to(dev()))

    def fresh(auto=False):
        m = GAOT(2, 1, model_cfg(model))
        m.load_state_dict(sd)
        m.to(dev()).train()
        m.auto_graph = auto
        return m

    ops.register_grad_slots([], [])
    m0 = fresh()
    ops.mse_loss(m0(pndata=p.to(dev()), **kw), t.to(dev())).backward()           # plain eager: no slots, no deferral
    ref = {k: q.grad.detach().clone() for k, q in m0.named_parameters()}
    top = max(float(v.norm()) for v in ref.values())
    err = lambda got: max(float((got[k] - ref[k]).norm()) / max(float(ref[k].norm()), 1e-3 * top) for k in ref)
    for graph in (False, True):
        m = fresh()
        ts = TrainStep(m, lr=0.0, weight_decay=0.0, use_graph=graph)
        ts.bind(p.to(dev()), t.to(dev()), **kw)
        if graph:
            ts.step(); ts.step()                                               # lr = 0: weights stay; the replay's gradients land in the flat buffer
        else:
            ts._forward_backward()
        torch.cuda.synchronize()
        view_of = {id(q_): v for q_, v in zip(ts.bucket.params, ts.bucket.views)}
        got = {k: view_of[id(q)].detach().clone() for k, q in m.named_parameters()}
        assert err(got) < 2e-6, ("TrainStep", graph, err(got))
        ts.bucket.clear()
        ops.register_grad_slots([], [])
    m = fresh(auto=True)                                                          # the reference's own loop, auto-graphed from step 2 on
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    for i in range(4):
        opt.zero_grad()
        torch.nn.functional.mse_loss(m(pndata=p.to(dev()), **kw), t.to(dev())).backward()
        got = {k: q.grad.detach().clone() for k, q in m.named_parameters()}
        assert err(got) < 2e-6, ("autograph", i, err(got))
    assert len(m._auto_graph_cache) == 1


# (autograph in vx mode -- any batch composition replays -- is pinned in tests/test_vx_static_gpu.py)


def test_c4_ns_gauss_16k_pair_step_and_10_step_rollout_vs_oracle():
    """BASELINE configs[3] at its stated single-GPU shape: fx, 16 384 nodes, batch 4, example model with in 4 / out 2."""
    import numpy as np
    from oracle import gaot_oracle as O
    model, sd, ocfg = make_model(4, 2, [64, 64], precompute=False, seed=9)
    ocfg = O.OracleConfig(**{**ocfg.__dict__, "precompute_edges": False})
    g = torch.Generator().manual_seed(9)
    lat, x = grid([64, 64]), uniform_points(16384, 2, g)
    p, tgt = torch.randn(4, 16384, 4, generator=g), torch.randn(4, 16384, 2, generator=g)
    enc, dec = [O.radius_csr(x, lat, 0.033, exact=True)], [O.radius_csr(lat, x, 0.033, exact=True)]
    ocfg_pre = O.OracleConfig(**{**ocfg.__dict__, "precompute_edges": True})
    loss, grads, _, _, pred = O.train_step(sd, ocfg_pre, dict(latent=lat, xcoord=x, pndata=p, target=tgt, encoder_nbrs=enc, decoder_nbrs=dec),
                                           return_pred=True)
    model.to(dev()).train()
    kw = dict(latent_tokens_coord=lat.to(dev()), xcoord=x.to(dev()))
    from gaot_amd import ops
    model.zero_grad(set_to_none=True)
    out = model(pndata=p.to(dev()), **kw)
    l = ops.mse_loss(out, tgt.to(dev()))
    l.backward()
    assert rel_l2(out.detach().cpu(), pred) < OUT_TOL and abs(float(l.detach()) - float(loss)) < LOSS_TOL * abs(float(loss))
    errs = grad_errors(model, grads, fp32_noise(sd, ocfg_pre, dict(latent=lat, xcoord=x, pndata=p, target=tgt, encoder_nbrs=enc, decoder_nbrs=dec), grads))
    gated = ("encoder.geoembed.mlp.0.weight", "encoder.geoembed.mlp.0.bias", "decoder.geoembed.mlp.0.weight", "decoder.geoembed.mlp.0.bias")
    assert max(v for k, v in errs.items() if k not in gated) < GRAD_TOL
    assert max(errs[k] for k in gated) < 1e-3          # statistics-gated tensors: the reference's own fp32 conditioning (see the C2 test)
    # 10-step rollout, stepper 'time_der'
    stats = {"u": {"mean": torch.tensor([0.1, -0.2]), "std": torch.tensor([1.5, 0.7])},
             "der": {"mean": torch.tensor([-0.05, 0.03]), "std": torch.tensor([0.8, 1.1])},
             "start_time": {"mean": 0.4, "std": 0.25}, "time_diffs": {"mean": 0.1, "std": 0.05}}
    tv, ti = np.linspace(0.0, 1.0, 21), np.arange(0, 22, 2)[:11]
    xb = torch.randn(4, 16384, 2, generator=g)
    ref = O.autoregressive_predict(sd, ocfg_pre, xb, ti, tv, stats, "time_der", lat, x, encoder_nbrs=enc, decoder_nbrs=dec)
    model.eval()
    got = model.autoregressive_predict(x_batch=xb.to(dev()), time_indices=ti, t_values=tv, stats=stats, stepper_mode="time_der",
                                       latent_tokens_coord=lat.to(dev()), fixed_coord=x.to(dev()))
    assert got.shape == (4, 10, 16384, 2)
    per_step = [rel_l2(got[:, i].cpu(), ref[:, i]) for i in range(10)]
    print("[C4 rollout] rel-L2 per step:", " ".join(f"{e:.1e}" for e in per_step))
    assert per_step[0] < OUT_TOL and max(per_step) < 5e-5


def test_attention_dropout_in_the_model_train_vs_eval():
    """atten_dropout > 0 (attn.py:110-114): active in training mode only, a new mask per call, also inside the captured
    training graphs (the seed lives on the device); evaluation equals the model without dropout"""
    from gaot_amd import ops
    from gaot_amd.model.gaot import GAOT
    from gaot_amd.model.layers.attn import AttentionConfig
    model, sd, _ = make_model(2, 1, [16, 16], C=32, hidden=128, heads=4, radius=0.12, precompute=False, seed=5)
    cfg = model_cfg(model)
    cfg.args.transformer.attn_config = AttentionConfig(num_heads=4, num_kv_heads=4, atten_dropout=0.2)
    drop = GAOT(2, 1, cfg); drop.load_state_dict(sd); drop.to(dev())
    model.to(dev())
    g = torch.Generator().manual_seed(5)
    lat, x = grid([16, 16]).to(dev()), uniform_points(700, 2, g).to(dev())
    p = torch.randn(2, 700, 2, generator=g).to(dev()); t = torch.randn(2, 700, 1, generator=g).to(dev())
    model.eval(); drop.eval()
    with torch.no_grad():
        assert torch.equal(drop(latent_tokens_coord=lat, xcoord=x, pndata=p), model(latent_tokens_coord=lat, xcoord=x, pndata=p))
    drop.train(); model.train()
    ops.seed_dropout(99, dev())
    outs, losses = [], []
    for step in range(4):            # the third and fourth calls replay captured graphs (autograph)
        drop.zero_grad(set_to_none=True)
        y = drop(latent_tokens_coord=lat, xcoord=x, pndata=p)
        loss = ops.mse_loss(y, t); loss.backward()
        assert all(torch.isfinite(q.grad).all() for q in drop.parameters() if q.grad is not None)
        outs.append(y.detach().clone()); losses.append(float(loss.detach()))
    ref = model(latent_tokens_coord=lat, xcoord=x, pndata=p).detach()
    for i in range(4):
        assert 1e-4 < rel_l2(outs[i].cpu(), ref.cpu()) < 0.5, rel_l2(outs[i].cpu(), ref.cpu())
        for j in range(i):
            assert not torch.equal(outs[i], outs[j])


def test_head_dim_128_model_vs_oracle():
    """hidden 256 on 2 heads: head_dim 128 (the fp32-MFMA attention kernels above 64) -- forward, loss and every gradient"""
    from oracle import gaot_oracle as O
    model, sd, ocfg = make_model(2, 1, [16, 16], C=32, hidden=256, heads=2, radius=0.12, seed=21)
    g = torch.Generator().manual_seed(21)
    lat, x = grid([16, 16]), uniform_points(900, 2, g)
    p, tgt = torch.randn(2, 900, 2, generator=g), torch.randn(2, 900, 1, generator=g)
    enc, dec = [O.radius_csr(x, lat, 0.12)], [O.radius_csr(lat, x, 0.12)]
    batch = dict(latent=lat, xcoord=x, pndata=p, target=tgt, encoder_nbrs=enc, decoder_nbrs=dec)
    loss, grads, _, _, pred = O.train_step(sd, ocfg, batch, return_pred=True)
    model.to(dev()).train()
    assert model.processor.blocks_in_order()[0].attn.head_dim == 128
    kw = dict(latent_tokens_coord=lat.to(dev()), xcoord=x.to(dev()), encoder_nbrs=[csr_dict(enc[0])], decoder_nbrs=[csr_dict(dec[0])])
    check_step(model, (loss, grads, pred), kw, p.to(dev()), tgt.to(dev()), "head_dim 128", noise=fp32_noise(sd, ocfg, batch, grads))


@pytest.mark.parametrize("graph", [False, True])
def test_twelve_training_steps_track_the_oracle(graph):
    """TrainStep (eager and hipGraph replay) against the oracle's loop over 12 optimizer steps on changing batches: the loss
    history and the final weights stay together (differences compound through AdamW's normalisation, hence the looser bars:
    loss 1e-4 relative, weights 1e-3 of the largest update -- an element whose gradient is at the rounding level moves by lr
    whatever its value)"""
    from gaot_amd.model.gaot import GAOT
    from gaot_amd.trainer import TrainStep
    from oracle import gaot_oracle as O
    model, sd, ocfg = make_model(2, 1, [16, 16], C=32, hidden=128, heads=4, radius=0.12, precompute=False, seed=31)
    g = torch.Generator().manual_seed(31)
    lat, x = grid([16, 16]), uniform_points(800, 2, g)
    data = [(torch.randn(3, 800, 2, generator=g), torch.randn(3, 800, 1, generator=g)) for _ in range(12)]
    enc, dec = [O.radius_csr(x, lat, 0.12, exact=True)], [O.radius_csr(lat, x, 0.12, exact=True)]
    cur, state, ref_losses = sd, None, []
    for p, t in data:
        loss, _, cur, state = O.train_step(cur, ocfg, dict(latent=lat, xcoord=x, pndata=p, target=t, encoder_nbrs=enc, decoder_nbrs=dec),
                                           lr=2e-3, weight_decay=1e-4, state=state)
        ref_losses.append(float(loss))
    m = GAOT(2, 1, model_cfg(model)); m.load_state_dict(sd); m.to(dev()).train()
    ts = TrainStep(m, lr=2e-3, weight_decay=1e-4, use_graph=graph)
    ts.bind(data[0][0].to(dev()), data[0][1].to(dev()), latent_tokens_coord=lat.to(dev()), xcoord=x.to(dev()))
    losses = []
    for p, t in data:
        losses.append(float(ts.step(p.to(dev()), t.to(dev()))))
    torch.cuda.synchronize()
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) < 1e-4 * abs(b), (losses, ref_losses)
    moved = max(float((cur[k] - sd[k]).abs().max()) for k in sd)
    worst = max(float((prm.detach().cpu() - cur[k]).abs().max()) for k, prm in m.state_dict().items())
    assert worst < 1e-3 * moved, (worst, moved)


@pytest.mark.parametrize("graph", [False, True])
def test_trainstep_one_launch_loss_equals_the_autograd_loss(graph):
    """TrainStep's loss / seed gradient / optimizer tick come from one launch (ops.mse_loss_and_grad, FlatAdamW.step(ticked=True));
    `fused_loss = False` keeps the three launches + the 1-thread tick.  Same losses, same weights, same step counter, bit for bit."""
    from gaot_amd.trainer import TrainStep
    from gaot_amd.model.gaot import GAOT
    model, sd, ocfg = make_model(2, 1, [16, 16], C=32, hidden=128, heads=4, radius=0.12, precompute=False, seed=32)
    g = torch.Generator().manual_seed(32)
    lat, x = grid([16, 16]), uniform_points(800, 2, g)
    data = [(torch.randn(3, 800, 2, generator=g), torch.randn(3, 800, 1, generator=g)) for _ in range(5)]
    out = []
    for fused in (True, False):
        m = GAOT(2, 1, model_cfg(model)); m.load_state_dict(sd); m.to(dev()).train()
        ts = TrainStep(m, lr=2e-3, weight_decay=1e-4, use_graph=graph)
        ts.fused_loss = fused
        ts.bind(data[0][0].to(dev()), data[0][1].to(dev()), latent_tokens_coord=lat.to(dev()), xcoord=x.to(dev()))
        losses = [ts.step(p.to(dev()), t.to(dev())).clone() for p, t in data]
        torch.cuda.synchronize()
        out.append((losses, [q.detach().clone() for q in m.parameters()], float(ts.opt.step_count)))
    assert out[0][2] == out[1][2] == 5.0
    assert all(torch.equal(a, b) for a, b in zip(out[0][0], out[1][0]))
    assert all(torch.equal(a, b) for a, b in zip(out[0][1], out[1][1]))


@pytest.mark.parametrize("graph", [True, False])
def test_two_trainings_from_one_seed_in_one_process_end_bit_identical(graph):
    """Every reduction on the path has a fixed order (slab sums, the query-split halves, the grouped launch's tickets), so a training
    is a function of its seed -- also the SECOND one in a process, whose parameters' id()s may reuse those of the first one's
    temporaries (tests/test_host_cpu.py: a stale id once took a parameter's gradient slot away, and with it the product's path).
    Two trainings as history, then two compared: 30 steps of the 4 096-token batch (BASELINE configs[3]'s token count), weights equal
    bit for bit, and every step's flat gradient buffer too."""
    import bench
    from gaot_amd import ops
    from gaot_amd.trainer import TrainStep

    def make(g):
        ops.register_grad_slots([], [])
        torch.manual_seed(0)
        model = bench.build_model().to(dev()).train()
        lat, x, p, t = bench.synthetic(1234, dev())
        ts = TrainStep(model, lr=8e-4, weight_decay=1e-5, use_graph=g)
        ts.bind(p[:4].contiguous(), t[:4].contiguous(), latent_tokens_coord=lat, xcoord=x)
        return ts, model

    for _ in range(2):                                   # history: allocator state and freed ids of earlier trainings
        ts, model = make(True)
        for _ in range(12):
            ts.step()
        torch.cuda.synchronize()
        del ts, model
    runs = []
    for _ in range(2):
        ts, model = make(graph)
        grads = []
        for _ in range(30):
            ts.step()
            grads.append(ts.bucket.flat.clone())
        torch.cuda.synchronize()
        runs.append((grads, [q.detach().clone() for q in model.parameters()]))
        del ts, model
    first = next((i for i, (a, b) in enumerate(zip(runs[0][0], runs[1][0])) if not torch.equal(a, b)), None)
    assert first is None, f"flat gradients differ from step {first} on"
    assert all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1]))
    ops.register_grad_slots([], [])
