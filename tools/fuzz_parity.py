"""Randomised parity sweep: HIP path against the CPU oracle on random configurations of the operator API (not part of the product;
`gpurun -- python tools/fuzz_parity.py <first seed> <count>`).  Every seed draws a MAGNO / transformer configuration, a mode (fx, fx with a
separate query cloud, vx with per-sample graphs), a small random geometry and a batch; the HIP model takes the oracle-initialised weights of
its own state_dict, runs forward + MSE + backward, and is compared with oracle.train_step on the same inputs under the bars of the GPU tests
(output / loss 1e-5, every gradient tensor within max(1e-4 of its norm, 3 x the reference's own fp32 rounding on it): tests/_golden.py).
One line per seed; exit code 1 if any seed fails."""
import json
import os
import random
import sys
import traceback
from types import SimpleNamespace as NS

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gaot_oracle as O                                              # noqa: E402  (the checker)
from tests._golden import fp32_noise, rel_l2, unfloored_ratio                    # noqa: E402

OUT_TOL, LOSS_TOL, GRAD_TOL = 1e-5, 1e-5, 1e-4


def draw(seed):
    r = random.Random(seed)
    d = r.choice([2, 2, 2, 3])
    P = r.choice([1, 2, 2, 2, 4]) if d == 2 else r.choice([1, 2, 2])
    if d == 2:
        sizes = r.choice([[8, 8], [16, 8], [16, 16], [8, 16]])
    else:
        sizes = r.choice([[4, 4, 4], [8, 4, 4], [4, 8, 4]])
    sizes = [max(s, P) for s in sizes]
    scales = r.choice([[1.0], [1.0], [0.5, 1.0], [1.0, 1.5]])
    heads = r.choice([4, 8])
    hidden = r.choice([64, 128, 256]) if d == 2 else r.choice([96, 128, 192, 384])
    transform = r.choice(["linear", "linear", "linear", "nonlinear", "nonlinear_kernelonly"])
    use_attention = r.random() < 0.75
    magno = dict(coord_dim=d, radius=(0.35 if d == 2 else 0.6) * r.uniform(0.8, 1.2), hidden_size=r.choice([32, 64]), mlp_layers=r.choice([2, 3]),
                 lifting_channels=r.choice([16, 32, 64]) if d == 2 else r.choice([24, 48]),      # (3-D: the reference's sinusoidal embedding of the P^3 C wide tokens needs a multiple of 6, gaot.py:119-130)
                 scales=scales, use_scale_weights=len(scales) > 1 and r.random() < 0.5,
                 use_attention=use_attention, attention_type=r.choice(["cosine", "cosine", "dot_product"]),
                 use_geoembed=r.random() < 0.8, embedding_method=r.choice(["statistical", "statistical", "pointnet"]),
                 pooling=r.choice(["max", "mean"]), transform_type=transform, node_embedding=r.random() < 0.25)
    tf = dict(patch_size=P, hidden_size=hidden, num_layers=r.choice([1, 2, 3, 4, 5]), positional_embedding=r.choice(["absolute", "absolute", "rope"]),
              use_long_range_skip=r.random() < 0.7, use_attn_norm=r.random() < 0.8, use_ffn_norm=r.random() < 0.8, ffn_multiplier=r.choice([2, 4]))
    attn = dict(num_heads=heads, num_kv_heads=r.choice([heads, heads, heads // 2]), use_conditional_norm=r.random() < 0.2)
    mode = r.choice(["fx", "fx", "fx_query", "vx", "fx_own_search"])
    cin = r.choice([1, 2, 3])
    if transform != "linear":
        cin = magno["lifting_channels"]          # the reference's encoder kernel takes 2 d + in_channels inputs but is fed the LIFTED features (magno.py:112-116): only equal widths run
    return NS(seed=seed, d=d, sizes=sizes, magno=magno, tf=tf, attn=attn, mode=mode, B=r.choice([1, 2, 3, 5]), N=r.randrange(150, 600),
              Nq=r.randrange(100, 400), cin=cin, cout=r.choice([1, 2]))


def oracle_config(c):
    m, t, a = c.magno, c.tf, c.attn
    return O.OracleConfig(coord_dim=c.d, radius=m["radius"], hidden_size=m["hidden_size"], mlp_layers=m["mlp_layers"],
                          lifting_channels=m["lifting_channels"], scales=m["scales"], use_scale_weights=m["use_scale_weights"],
                          use_attention=m["use_attention"], attention_type=m["attention_type"], use_geoembed=m["use_geoembed"],
                          embedding_method=m["embedding_method"], pooling=m["pooling"], transform_type=m["transform_type"],
                          node_embedding=m["node_embedding"], precompute_edges=True, patch_size=t["patch_size"], tf_hidden_size=t["hidden_size"],
                          use_attn_norm=t["use_attn_norm"], use_ffn_norm=t["use_ffn_norm"], num_layers=t["num_layers"],
                          positional_embedding=t["positional_embedding"], use_long_range_skip=t["use_long_range_skip"],
                          ffn_multiplier=t["ffn_multiplier"], num_heads=a["num_heads"], num_kv_heads=a["num_kv_heads"],
                          use_conditional_norm=a["use_conditional_norm"], latent_tokens_size=c.sizes)


def make_batch(c):
    """the seed's geometry, fields and exact radius graphs (CPU tensors, the oracle's batch layout)"""
    g = torch.Generator().manual_seed(1000 + c.seed)
    m = c.magno
    lat = O.latent_grid(c.sizes)
    vx = c.mode == "vx"
    x = (torch.rand(c.B, c.N, c.d, generator=g) if vx else torch.rand(c.N, c.d, generator=g)) * 2 - 1
    q = torch.rand(c.Nq, c.d, generator=g) * 2 - 1 if c.mode == "fx_query" else None
    Nout = c.Nq if q is not None else c.N
    p, tgt = torch.randn(c.B, c.N, c.cin, generator=g), torch.randn(c.B, Nout, c.cout, generator=g)
    cond = torch.rand(c.B, 1, generator=g) if c.attn["use_conditional_norm"] else None
    rs = [m["radius"] * s for s in m["scales"]]
    if vx:
        enc = [[O.radius_csr(x[b], lat, r, exact=True) for r in rs] for b in range(c.B)]
        dec = [[O.radius_csr(lat, x[b], r, exact=True) for r in rs] for b in range(c.B)]
    else:
        enc = [O.radius_csr(x, lat, r, exact=True) for r in rs]
        dec = [O.radius_csr(lat, x if q is None else q, r, exact=True) for r in rs]
    batch = dict(latent=lat, xcoord=x, pndata=p, target=tgt, encoder_nbrs=enc, decoder_nbrs=dec)
    if q is not None:
        batch["query_coord"] = q
    if cond is not None:
        batch["condition"] = cond
    return batch


def run(c, dev):
    from gaot_amd.model.gaot import GAOT
    from gaot_amd.model.layers.attn import AttentionConfig, TransformerConfig
    from gaot_amd.model.layers.magno import MAGNOConfig
    from gaot_amd import ops
    torch.manual_seed(c.seed)
    pre = c.mode != "fx_own_search"
    model = GAOT(c.cin, c.cout, NS(args=NS(magno=MAGNOConfig(precompute_edges=pre, **c.magno),
                                           transformer=TransformerConfig(attn_config=AttentionConfig(**c.attn), **c.tf)),
                                   latent_tokens_size=c.sizes))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    m, t, a = c.magno, c.tf, c.attn
    ocfg = O.OracleConfig(coord_dim=c.d, radius=m["radius"], hidden_size=m["hidden_size"], mlp_layers=m["mlp_layers"],
                          lifting_channels=m["lifting_channels"], scales=m["scales"], use_scale_weights=m["use_scale_weights"],
                          use_attention=m["use_attention"], attention_type=m["attention_type"], use_geoembed=m["use_geoembed"],
                          embedding_method=m["embedding_method"], pooling=m["pooling"], transform_type=m["transform_type"],
                          node_embedding=m["node_embedding"], precompute_edges=True, patch_size=t["patch_size"], tf_hidden_size=t["hidden_size"],
                          use_attn_norm=t["use_attn_norm"], use_ffn_norm=t["use_ffn_norm"], num_layers=t["num_layers"],
                          positional_embedding=t["positional_embedding"], use_long_range_skip=t["use_long_range_skip"],
                          ffn_multiplier=t["ffn_multiplier"], num_heads=a["num_heads"], num_kv_heads=a["num_kv_heads"],
                          use_conditional_norm=a["use_conditional_norm"], latent_tokens_size=c.sizes)
    batch = make_batch(c)
    lat, x, p, tgt, enc, dec = (batch[k] for k in ("latent", "xcoord", "pndata", "target", "encoder_nbrs", "decoder_nbrs"))
    q, cond, vx = batch.get("query_coord"), batch.get("condition"), c.mode == "vx"
    loss_ref, grads_ref, _, _, pred_ref = O.train_step(sd, ocfg, batch, return_pred=True)
    noise = fp32_noise(sd, ocfg, batch, grads_ref)
    if os.environ.get("FUZZ_ORACLE_ONLY"):          # (CPU dry run of the checker's half)
        return True, dict(loss_ref=float(loss_ref))
    model = model.to(dev).train()
    one = lambda cs: {"neighbors_index": cs[0].to(dev), "neighbors_row_splits": cs[1].to(dev)}
    kw = dict(latent_tokens_coord=lat.to(dev), xcoord=x.to(dev))
    if pre:
        kw.update(encoder_nbrs=[[one(s) for s in row] for row in enc] if vx else [one(s) for s in enc],
                  decoder_nbrs=[[one(s) for s in row] for row in dec] if vx else [one(s) for s in dec])
    if q is not None:
        kw["query_coord"] = q.to(dev)
    if cond is not None:
        kw["condition"] = cond.to(dev)
    pred = model(pndata=p.to(dev), **kw)
    loss = ops.mse_loss(pred, tgt.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    e_out = rel_l2(pred.detach().cpu(), pred_ref)
    e_loss = abs(float(loss.detach()) - float(loss_ref)) / abs(float(loss_ref))
    out_bar = OUT_TOL
    if e_out >= OUT_TOL:
        # the reference's own fp32 rounding on the prediction can exceed 1e-5 (seed 649: five blocks without an attention norm; the fp32 oracle
        # is 1.2e-5 off its float64 evaluation, the HIP path 6.4e-7): the prediction is then held to the float64 oracle at 1e-5 instead
        dbl = lambda v: v.double() if torch.is_tensor(v) and v.is_floating_point() else v
        pred64 = O.train_step({k: dbl(v) for k, v in sd.items()}, ocfg, {k: dbl(v) for k, v in batch.items()}, return_pred=True)[4]
        if rel_l2(pred_ref.double(), pred64) > 0.5 * OUT_TOL:
            e_out = rel_l2(pred.detach().cpu().double(), pred64)
    got = {k: prm.grad for k, prm in model.named_parameters()}
    errs = {k: GRAD_TOL * v for k, v in unfloored_ratio(got, {k: grads_ref[k] for k in got}, noise, GRAD_TOL).items()}
    # a tensor that is ZERO in exact arithmetic (a key bias under a softmax: the fp32 reference's value is its own rounding, |g32| ~ |g32 - g64|)
    # has no scale of its own: the kernels' rounding there is held to the tolerance of its module's weight gradient instead of 3 x the
    # reference's (whose softmax backward happens to cancel better: seed 69, decoder.agno.key_proj.bias 1.5e-10 against a weight gradient of 3.3e-6)
    zero_exact = []
    for k in [k for k, v in errs.items() if v >= GRAD_TOL]:
        sib = k.rsplit(".", 1)[0] + ".weight"
        if sib != k and sib in grads_ref and float(grads_ref[k].double().norm()) <= 1.5 * noise[k]:
            err = float((got[k].detach().cpu().double() - grads_ref[k].double()).norm())
            if err <= GRAD_TOL * float(grads_ref[sib].double().norm()):
                errs[k] = GRAD_TOL * err / (GRAD_TOL * float(grads_ref[sib].double().norm()))
                zero_exact.append(k)
    worst = max(errs, key=errs.get)
    # gradients of the geometry-embedding MLP are CONDITIONED on ReLU gates that sit right behind the standardised statistics: one pre-activation
    # within 1e-8 of zero lands on either side depending on the last bit of a statistic (seed 203: |z| = 1.0e-8 in mlp.2, one flipped gate of
    # 40 960 moves mlp.0.weight's gradient by 2.4e-3 between two float64 evaluations of the chain that differ only in whose fp32 statistics
    # they start from; tools history: DESIGN section 2).  Such seeds are reported as `stats_gate` (everything else of the seed must pass) and do
    # not fail the sweep; tests/test_configs_gpu.py pins the same effect on the bench configuration (c2_stats_gates.npz).
    failing = [k for k, v in errs.items() if v >= GRAD_TOL]
    stats_gate = bool(failing) and all(".geoembed." in k for k in failing)      # (the pointnet branch has the same ReLU gates behind its per-edge chain: seed 264)
    if stats_gate:
        rest = {k: v for k, v in errs.items() if k not in failing}
        worst = max(rest, key=rest.get)
    ok = e_out < OUT_TOL and e_loss < LOSS_TOL and errs[worst] < GRAD_TOL
    if os.environ.get("FUZZ_DUMP"):          # the reference's own fp32 rounding on the prediction, the eight worst tensors of the seed with their norms
        dbl = lambda v: v.double() if torch.is_tensor(v) and v.is_floating_point() else v
        pred64 = O.train_step({k: dbl(v) for k, v in sd.items()}, ocfg, {k: dbl(v) for k, v in batch.items()}, return_pred=True)[4]
        print(json.dumps({"prediction": {"hip_vs_fp32_oracle": e_out, "hip_vs_fp64_oracle": rel_l2(pred.detach().cpu().double(), pred64),
                                         "fp32_oracle_vs_fp64_oracle": rel_l2(pred_ref.double(), pred64), "absmax": float(pred_ref.abs().max())}}), flush=True)
        for k in sorted(errs, key=errs.get, reverse=True)[:8]:
            print(json.dumps({"tensor": k, "figure": errs[k], "norm": float(grads_ref[k].norm()), "noise": noise[k],
                              "err": float((got[k].detach().cpu().double() - grads_ref[k].double()).norm()) if got[k] is not None else None}), flush=True)
    info = dict(stats_gate=failing if stats_gate else [], out=e_out, loss=e_loss, grad=errs[worst], worst=worst, zero_in_exact_arithmetic=zero_exact, edges=int(sum(e[0].numel() for e in (enc if not vx else enc[0]))))
    # the same step four more times as the unchanged reference loop issues it (zero_grad / forward / nn.MSELoss / backward, no optimizer step):
    # where autograph serves the shapes (fx with the module's own graphs, vx) the later iterations are hipGraph replays -- same weights, so
    # prediction and gradients must stay where the first (eager) pass put them
    first = {k: (v.detach().clone() if v is not None else None) for k, v in got.items()}
    pred0 = pred.detach().clone()
    lossf = torch.nn.MSELoss()
    pd_, td_ = p.to(dev), tgt.to(dev)
    for _ in range(4):
        model.zero_grad(set_to_none=True)
        pk = model(pndata=pd_, **kw)
        lossf(pk, td_).backward()
    torch.cuda.synchronize()
    rep_out = float((pk.detach() - pred0).norm() / pred0.norm())
    rep_grad = 0.0
    top = max(float(v.norm()) for v in first.values() if v is not None)
    for k, prm in model.named_parameters():
        a, b = prm.grad, first[k]
        if (a is None) != (b is None):
            rep_grad = float("inf")
        elif a is not None:
            rep_grad = max(rep_grad, float((a - b).norm()) / max(float(b.norm()), 1e-6 * top))
    # inference passes (eval, no_grad; the second one is served from the inference caches) against the oracle's prediction
    model.eval()
    with torch.no_grad():
        model(pndata=pd_, **kw)
        pe = model(pndata=pd_, **kw)
    e_eval = rel_l2(pe.cpu(), pred_ref)
    info.update(replay_out=rep_out, replay_grad=rep_grad, eval_out=e_eval)
    ok = ok and rep_out < 2e-6 and rep_grad < 2e-5 and e_eval < OUT_TOL
    return ok, info


def rollout_setup(c):
    """the rollout case run_rollout draws for a configuration (same draws, CPU tensors): what the reference-side generator
    (tests/golden/make_fuzz_reference_rollouts.py) and the oracle-side test share"""
    import numpy as np
    r = random.Random(7000 + c.seed)
    g = torch.Generator().manual_seed(5000 + c.seed)
    udim, cdim = c.cout, r.choice([0, 1])
    cn = c.attn["use_conditional_norm"]
    cin = udim + cdim + (1 if cn else 2)
    lat = O.latent_grid(c.sizes)
    x = torch.rand(c.N, c.d, generator=g) * 2 - 1
    rs = [c.magno["radius"] * s for s in c.magno["scales"]]
    enc = [O.radius_csr(x, lat, rad, exact=True) for rad in rs]
    dec = [O.radius_csr(lat, x, rad, exact=True) for rad in rs]
    steps = r.choice([3, 4, 6])
    mode = r.choice(["output", "residual", "time_der"])
    vec = lambda n, lo, hi: torch.tensor([r.uniform(lo, hi) for _ in range(n)])
    stats = {"u": {"mean": vec(udim, -0.3, 0.3), "std": vec(udim, 0.6, 1.6)}, "res": {"mean": vec(udim, -0.1, 0.1), "std": vec(udim, 0.5, 1.2)},
             "der": {"mean": vec(udim, -0.1, 0.1), "std": vec(udim, 0.5, 1.2)}, "start_time": {"mean": 0.4, "std": 0.25},
             "time_diffs": {"mean": 0.1, "std": 0.05}}
    if cdim:
        stats["c"] = {"mean": vec(cdim, -0.1, 0.1), "std": vec(cdim, 0.8, 1.2)}
    tv = np.linspace(0.0, 1.0, 4 * steps + 1)
    ti = np.arange(0, 2 * steps + 2, 2)[:steps + 1]
    xb = torch.randn(c.B, c.N, udim + cdim, generator=g)
    return NS(udim=udim, cdim=cdim, cn=cn, cin=cin, lat=lat, x=x, enc=enc, dec=dec, steps=steps, mode=mode, stats=stats, tv=tv, ti=ti, xb=xb)


def run_rollout(c, dev):
    """autoregressive_predict (gaot.py:307-476) of a random fx configuration against the oracle's: 3-6 steps, a random stepper mode, +- one
    constant channel, +- conditional norm; udim = c.cout state channels, the model's input = state (+ constant) + the two time columns
    (conditional norm: the last time column is the condition instead, gaot.py:403-408)."""
    import numpy as np
    from gaot_amd.model.gaot import GAOT
    from gaot_amd.model.layers.attn import AttentionConfig, TransformerConfig
    from gaot_amd.model.layers.magno import MAGNOConfig
    r = random.Random(7000 + c.seed)
    g = torch.Generator().manual_seed(5000 + c.seed)
    torch.manual_seed(c.seed)
    udim, cdim = c.cout, r.choice([0, 1])
    cn = c.attn["use_conditional_norm"]
    cin = udim + cdim + (1 if cn else 2)
    pre = c.mode != "fx_own_search"
    model = GAOT(cin, udim, NS(args=NS(magno=MAGNOConfig(precompute_edges=pre, **c.magno),
                                       transformer=TransformerConfig(attn_config=AttentionConfig(**c.attn), **c.tf)), latent_tokens_size=c.sizes))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    m, t, a = c.magno, c.tf, c.attn
    ocfg = O.OracleConfig(coord_dim=c.d, radius=m["radius"], hidden_size=m["hidden_size"], mlp_layers=m["mlp_layers"],
                          lifting_channels=m["lifting_channels"], scales=m["scales"], use_scale_weights=m["use_scale_weights"],
                          use_attention=m["use_attention"], attention_type=m["attention_type"], use_geoembed=m["use_geoembed"],
                          embedding_method=m["embedding_method"], pooling=m["pooling"], transform_type=m["transform_type"],
                          node_embedding=m["node_embedding"], precompute_edges=True, patch_size=t["patch_size"], tf_hidden_size=t["hidden_size"],
                          use_attn_norm=t["use_attn_norm"], use_ffn_norm=t["use_ffn_norm"], num_layers=t["num_layers"],
                          positional_embedding=t["positional_embedding"], use_long_range_skip=t["use_long_range_skip"],
                          ffn_multiplier=t["ffn_multiplier"], num_heads=a["num_heads"], num_kv_heads=a["num_kv_heads"],
                          use_conditional_norm=cn, latent_tokens_size=c.sizes)
    lat = O.latent_grid(c.sizes)
    x = torch.rand(c.N, c.d, generator=g) * 2 - 1
    rs = [m["radius"] * s for s in m["scales"]]
    enc = [O.radius_csr(x, lat, rad, exact=True) for rad in rs]
    dec = [O.radius_csr(lat, x, rad, exact=True) for rad in rs]
    steps = r.choice([3, 4, 6])
    mode = r.choice(["output", "residual", "time_der"])
    vec = lambda n, lo, hi: torch.tensor([r.uniform(lo, hi) for _ in range(n)])
    stats = {"u": {"mean": vec(udim, -0.3, 0.3), "std": vec(udim, 0.6, 1.6)}, "res": {"mean": vec(udim, -0.1, 0.1), "std": vec(udim, 0.5, 1.2)},
             "der": {"mean": vec(udim, -0.1, 0.1), "std": vec(udim, 0.5, 1.2)}, "start_time": {"mean": 0.4, "std": 0.25},
             "time_diffs": {"mean": 0.1, "std": 0.05}}
    if cdim:
        stats["c"] = {"mean": vec(cdim, -0.1, 0.1), "std": vec(cdim, 0.8, 1.2)}
    tv = np.linspace(0.0, 1.0, 4 * steps + 1)
    ti = np.arange(0, 2 * steps + 2, 2)[:steps + 1]
    xb = torch.randn(c.B, c.N, udim + cdim, generator=g)
    ref = O.autoregressive_predict(sd, ocfg, xb, ti, tv, stats, mode, lat, x, use_conditional_norm=cn, encoder_nbrs=enc, decoder_nbrs=dec)
    model = model.to(dev).eval()
    one = lambda cs: {"neighbors_index": cs[0].to(dev), "neighbors_row_splits": cs[1].to(dev)}
    kw = dict(encoder_nbrs=[one(s_) for s_ in enc], decoder_nbrs=[one(s_) for s_ in dec]) if pre else {}
    got = model.autoregressive_predict(x_batch=xb.to(dev), time_indices=ti, t_values=tv, stats=stats, stepper_mode=mode,
                                       latent_tokens_coord=lat.to(dev), fixed_coord=x.to(dev), use_conditional_norm=cn, **kw)
    per = [rel_l2(got[:, i].cpu(), ref[:, i]) for i in range(steps)]
    return per[0] < OUT_TOL and max(per) < 5e-5, dict(rollout_mode=mode, rollout_steps=steps, rollout_first=per[0], rollout_worst=max(per))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = torch.device("cuda:0")
    bad = 0
    for seed in range(first, first + count):
        c = draw(seed)
        for k, v in json.loads(os.environ.get("FUZZ_OVERRIDE", "{}")).items():          # bisecting a seed: {"pooling": "max", "B": 1, ...}
            tgt = c.magno if k in c.magno else c.tf if k in c.tf else c.attn if k in c.attn else None
            if tgt is not None:
                tgt[k] = v
            else:
                setattr(c, k, v)
        try:
            ok, info = run(c, dev)
        except Exception as e:          # an unsupported combination must raise the reference's error, not crash: printed for a look
            ok, info = False, {"exception": f"{type(e).__name__}: {e}"[:400], "trace": traceback.format_exc().splitlines()[-6:]}
        if ok and c.mode in ("fx", "fx_own_search") and c.magno["transform_type"] == "linear" and not os.environ.get("FUZZ_ORACLE_ONLY"):
            try:
                ok2, info2 = run_rollout(c, dev)
            except Exception as e:
                ok2, info2 = False, {"rollout_exception": f"{type(e).__name__}: {e}"[:400], "trace": traceback.format_exc().splitlines()[-6:]}
            ok = ok and ok2
            info.update(info2)
        bad += 0 if ok else 1
        cfg = {**c.magno, **c.tf, **c.attn, "mode": c.mode, "sizes": c.sizes, "B": c.B, "N": c.N, "cin": c.cin, "cout": c.cout}
        print(json.dumps({"seed": seed, "ok": ok, **info, **({} if ok else {"config": cfg})}), flush=True)
    print(json.dumps({"seeds": count, "failed": bad}), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
