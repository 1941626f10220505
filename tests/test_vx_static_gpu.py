"""-m gpu: vx training on static padded unions (gaot_amd/plan.py StaticUnion, csrc/gno.hip union_compose_kernel).

The reference trains vx datasets under a shuffling loader (data_utils.py:272-294 collate, static_trainer.py:180-202): every step brings another
composition of per-sample graphs.  Here the batch's block-diagonal unions live in static buffers padded to an edge-count bucket and are composed
on the device from a table of per-sample plan pointers, so ONE captured step per bucket replays for any composition.  Pinned below:
  * the composed arrays equal the union composed by concatenation (plan.compose_plans) on the real edges, whatever was in the buffers before;
    the pads are harmless (no row references them, their edge scale is exactly 0);
  * model forward / loss / gradients on the static unions equal the composed-union path and the oracle (which loops over samples as the
    reference does, magno.py:356-413) on the permuted batch;
  * TrainStep: 8 shuffled steps replayed as hipGraphs equal 8 eager steps BIT FOR BIT (same kernels, same addresses) and track the oracle;
  * the unchanged reference loop (autograph) replays shuffled compositions and follows edited coordinates.
"""
import itertools

import pytest
import torch

from tests._golden import rel_l2
from tests._workloads import grid, naca_points
from tests.test_configs_gpu import GRAD_TOL, LOSS_TOL, OUT_TOL, check_step, csr_dict, dev, grad_errors, make_model

pytestmark = pytest.mark.gpu


def _dataset(n_samples, N, seed=0, lat_sizes=(64, 64), radius=0.033, spread=0.15, radius_jitter=0.0):
    """per-sample NACA-shaped meshes with their encoder / decoder radius graphs (oracle search = the reference's).  On a uniform latent grid a
    mesh of N points has ~N pi r^2 rho_latent edges whatever its shape, so batches of one dataset share one edge bucket; `radius_jitter` builds
    the graphs of sample i with radius * (1 + jitter * (i % 3)) -- caller-supplied lists may be anything -- to spread the batches over buckets."""
    from oracle import gaot_oracle as O
    g = torch.Generator().manual_seed(seed)
    lat = grid(list(lat_sizes))
    xs = [naca_points(N, g, spread) for _ in range(n_samples)]
    rs = [radius * (1.0 + radius_jitter * (i % 3)) for i in range(n_samples)]
    enc = [[O.radius_csr(x, lat, r)] for x, r in zip(xs, rs)]
    dec = [[O.radius_csr(lat, x, r)] for x, r in zip(xs, rs)]
    return lat, xs, enc, dec


def test_static_union_equals_composed_union_and_pads_are_inert():
    from gaot_amd import plan as P
    lat, xs, enc, dec = _dataset(6, 2048, seed=3, lat_sizes=(32, 32), radius=0.066)
    latd = lat.to(dev())
    xd = [x.to(dev()) for x in xs]
    B = 3
    for side in ("enc", "dec"):
        dicts = [csr_dict((enc if side == "enc" else dec)[i][0]) for i in range(6)]
        src_of = (lambda i: xd[i]) if side == "enc" else (lambda i: latd)
        dst_of = (lambda i: latd) if side == "enc" else (lambda i: xd[i])
        n_src, n_dst = src_of(0).shape[0], dst_of(0).shape[0]
        plans = [P.plan_for(d, n_src) for d in dicts]
        cap = P.edge_bucket(max(sum(plans[i].E for i in c) for c in itertools.combinations(range(6), B)))
        su = P.StaticUnion(B, n_src, n_dst, 2, 2, cap, dev())
        # poison the static buffers: nothing of an earlier batch (or of the allocation) may survive a refresh
        for t in (su.plan.index, su.plan.edge_query, su.plan.t_edge, su.plan.splits, su.plan.t_splits):
            t.fill_(-7)
        for order in ([5, 0, 3], [1, 2, 4], [3, 3, 0], [2, 1, 0]):          # big batch first, smaller later (stale tails), a repeated sample
            x_par = torch.stack([xd[i] for i in order])
            src_par, dst_par = (x_par, latd) if side == "enc" else (latd, x_par)
            su.load([plans[i] for i in order], src_par, dst_par)
            su.refresh()
            ref = P.MergedGeometry([dicts[i] for i in order], [src_of(i) for i in order], [dst_of(i) for i in order], build_parts=True)
            a, b = su.plan, ref.plan
            E = b.E
            assert int(a.e_dev.item()) == E == su.e_real and E <= a.E == cap
            for name in ("index", "edge_query", "t_edge"):
                assert torch.equal(getattr(a, name)[:E], getattr(b, name)[:E]), (side, name)
            assert torch.equal(a.splits, b.splits) and torch.equal(a.t_splits, b.t_splits)
            assert torch.equal(su.src, ref.src) and torch.equal(su.dst, ref.dst)
            # pads: valid indices, own ids in the transposed list
            assert bool((a.index[E:] == 0).all()) and bool((a.edge_query[E:] == 0).all())
            assert torch.equal(a.t_edge[E:], torch.arange(E, cap, device=dev(), dtype=torch.int32))
            assert torch.equal(a.edge_features(su.src, su.dst)[:E], b.edge_features(ref.src, ref.dst))
            cos = a.cosine_attention(su.src, su.dst)
            assert torch.equal(cos[:E], b.cosine_attention(ref.src, ref.dst)[:E]) and bool((cos[E:] == 0).all())
            assert torch.equal(su.geo_stats(), ref.geo_stats())
            inv = a.inv_deg_edge
            assert torch.equal(inv[:E], b.inv_deg_edge[:E]) and bool((inv[E:] == 0).all())
            assert torch.equal(a.deg, b.deg)
            # the same batch described by its raw int64 lists (dicts the trainer uploads per step): same CSR, same derived arrays; the
            # transposed CSR is derived on the device on request
            su.load_raw([dicts[i] for i in order], src_par, dst_par)
            su.refresh()
            assert int(a.e_dev.item()) == E
            for name in ("index", "edge_query", "t_edge"):
                assert torch.equal(getattr(a, name)[:E], getattr(b, name)[:E]), (side, "raw", name)
            assert torch.equal(a.splits, b.splits) and torch.equal(a.t_splits, b.t_splits)
            assert torch.equal(a.t_edge[E:], torch.arange(E, cap, device=dev(), dtype=torch.int32))
            assert torch.equal(su.geo_stats(), ref.geo_stats())
            assert torch.equal(a.cosine_attention(su.src, su.dst)[:E], b.cosine_attention(ref.src, ref.dst)[:E])
            assert int(su.flag.item()) == 0
        with pytest.raises(ValueError):
            su.load([plans[0]] * (B + 1), latd, latd)
        # a broken list is flagged on the device (and what is stored is clamped)
        bad = dict(dicts[0])
        bad["neighbors_index"] = dicts[0]["neighbors_index"].clone()
        bad["neighbors_index"][3] = n_src + 5
        su.load_raw([bad, dicts[1], dicts[2]], (torch.stack([xd[0], xd[1], xd[2]]) if side == "enc" else latd), (latd if side == "enc" else torch.stack([xd[0], xd[1], xd[2]])))
        su.refresh()
        assert int(su.flag.item()) == 2 and int(su.plan.index.max()) < B * n_src


@pytest.mark.parametrize("variant", ["default", "no_attention", "no_geoembed", "dot_product", "multiscale", "pointnet", "pointnet_mean", "kernelonly",
                                     "nonlinear", "nonlinear_kernelonly"])
def test_vx_static_path_equals_composed_path_and_oracle(variant):
    """forward, loss and every gradient of a vx batch: static padded unions vs composed unions (GAOT_VX_STATIC=0 path) vs the oracle"""
    from gaot_amd import plan as P
    from gaot_amd import ops
    from oracle import gaot_oracle as O
    kw = {"default": {}, "no_attention": dict(use_attention=False), "no_geoembed": dict(use_geoembed=False),
          "dot_product": dict(attention_type="dot_product"), "multiscale": dict(scales=[1.0, 0.5], use_scale_weights=True),
          "pointnet": dict(embedding_method="pointnet"), "pointnet_mean": dict(embedding_method="pointnet", pooling="mean"),
          "kernelonly": dict(transform_type="linear_kernelonly"), "nonlinear": dict(transform_type="nonlinear"),
          "nonlinear_kernelonly": dict(transform_type="nonlinear_kernelonly")}[variant]
    B, N = 3, 2048
    # ('nonlinear' kernels see f(y_j): the reference sizes their input by in_channels, magno.py:77-80, so lifting_channels == in_channels there)
    cin = 8 if variant.startswith("nonlinear") else 3
    model, sd, ocfg = make_model(cin, 1, [32, 32], radius=0.066, seed=11, **({"C": 8} if cin == 8 else {}), **kw)
    lat, xs, _, _ = _dataset(B, N, seed=5, lat_sizes=(32, 32), radius=0.066)
    scales = kw.get("scales", [1.0])
    enc = [[O.radius_csr(x, lat, 0.066 * s) for s in scales] for x in xs]
    dec = [[O.radius_csr(lat, x, 0.066 * s) for s in scales] for x in xs]
    g = torch.Generator().manual_seed(2)
    p, tgt = torch.randn(B, N, cin, generator=g), torch.randn(B, N, 1, generator=g)
    x = torch.stack(xs)
    model.to(dev()).train()
    fk = dict(latent_tokens_coord=lat.to(dev()), xcoord=x.to(dev()),
              encoder_nbrs=[[csr_dict(c) for c in row] for row in enc], decoder_nbrs=[[csr_dict(c) for c in row] for row in dec])
    res = {}
    for static in (True, False):
        P.VX_STATIC = static
        try:
            for rep in range(2):          # twice: the second pass reuses the static buffers
                model.zero_grad(set_to_none=True)
                pred = model(pndata=p.to(dev()), **fk)
                loss = ops.mse_loss(pred, tgt.to(dev()))
                loss.backward()
            torch.cuda.synchronize()
            res[static] = (pred.detach().cpu(), float(loss), {k: q.grad.detach().cpu().clone() for k, q in model.named_parameters() if q.grad is not None})
        finally:
            P.VX_STATIC = True
    assert len(model.encoder._static_unions) == len(scales) and len(model.decoder._static_unions) == len(scales)
    (ya, la, ga), (yb, lb, gb) = res[True], res[False]
    assert rel_l2(ya, yb) < 2e-6 and abs(la - lb) < 1e-6 * abs(lb)
    top = max(float(v.double().norm()) for v in gb.values())
    for k in gb:
        assert float((ga[k].double() - gb[k].double()).norm()) <= 2e-5 * max(float(gb[k].double().norm()), 1e-3 * top), k
    if variant == "default":
        lo, go, _, _, po = O.train_step(sd, ocfg, dict(latent=lat, xcoord=x, pndata=p, target=tgt, encoder_nbrs=enc, decoder_nbrs=dec), return_pred=True)
        assert rel_l2(ya, po) < OUT_TOL and abs(la - float(lo)) < LOSS_TOL * abs(float(lo))
        from tests._golden import fp32_noise, unfloored_ratio
        batch = dict(latent=lat, xcoord=x, pndata=p, target=tgt, encoder_nbrs=enc, decoder_nbrs=dec)
        ratio = unfloored_ratio(ga, {k: go[k] for k in ga}, fp32_noise(sd, ocfg, batch, go), GRAD_TOL)
        assert max(ratio.values()) <= 1.0, max(ratio, key=ratio.get)


def _shuffled_batches(n_samples, B, steps, seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randperm(n_samples, generator=g)[:B].tolist() for _ in range(steps)]


def test_trainstep_vx_shuffled_replay_equals_eager_bit_for_bit_and_tracks_the_oracle():
    """8 steps, every one a different composition drawn from a 10-sample dataset (edge totals spread over several buckets): TrainStep with
    hipGraph replay == TrainStep eager, bit for bit (losses and final weights); the first steps' losses equal the oracle's on the same batches."""
    from gaot_amd.model.gaot import GAOT
    from gaot_amd.trainer import TrainStep
    from oracle import gaot_oracle as O
    nS, B, N, steps = 10, 4, 2048, 8
    model, sd, ocfg = make_model(3, 1, [32, 32], radius=0.066, seed=4)
    lat, xs, enc, dec = _dataset(nS, N, seed=9, lat_sizes=(32, 32), radius=0.066, radius_jitter=0.12)
    # graphs of three different radii: the edge totals of the batches differ by more than one bucket
    g = torch.Generator().manual_seed(1)
    P_all, T_all = torch.randn(nS, N, 3, generator=g), torch.randn(nS, N, 1, generator=g)
    batches = _shuffled_batches(nS, B, steps, seed=21)
    latd = lat.to(dev())
    xd = torch.stack(xs).to(dev())
    encd = [[csr_dict(c) for c in row] for row in enc]
    decd = [[csr_dict(c) for c in row] for row in dec]
    runs = {}
    for graph in (True, False):
        m = GAOT(3, 1, _cfg_of(model))
        m.load_state_dict(sd)
        m.to(dev()).train()
        ts = TrainStep(m, lr=2e-3, weight_decay=1e-4, use_graph=graph)
        b0 = batches[0]
        ts.bind(P_all[b0].to(dev()), T_all[b0].to(dev()), latent_tokens_coord=latd, xcoord=xd[b0], encoder_nbrs=[encd[i] for i in b0],
                decoder_nbrs=[decd[i] for i in b0])
        losses = []
        for b in batches:
            losses.append(ts.step(P_all[b].to(dev()), T_all[b].to(dev()), xcoord=xd[b], encoder_nbrs=[encd[i] for i in b],
                                  decoder_nbrs=[decd[i] for i in b]).clone())
        torch.cuda.synchronize()
        runs[graph] = (torch.stack(losses).cpu(), torch.cat([q.detach().reshape(-1) for q in m.parameters()]).cpu(), ts)
    (lg, wg, tsg), (le, we, _) = runs[True], runs[False]
    assert torch.equal(lg, le), (lg, le)
    assert torch.equal(wg, we)
    n_sets = len(tsg._graph_sets)
    totals = sorted({sum(int(enc[i][0][0].numel()) for i in b) for b in batches})
    print(f"[vx shuffled] {steps} compositions, encoder edge totals {totals[0]}..{totals[-1]}, {n_sets} captured step(s)")
    assert 2 <= n_sets <= TrainStep.MAX_GRAPH_SETS and all(v["graphs"] is not None for v in tsg._graph_sets.values())
    # the oracle on the same sequence of batches (the reference loops over the samples of each batch)
    w = {k: v.clone() for k, v in sd.items()}
    mom = None
    for i, b in enumerate(batches[:3]):
        batch = dict(latent=lat, xcoord=torch.stack([xs[j] for j in b]), pndata=P_all[b], target=T_all[b], encoder_nbrs=[enc[j] for j in b],
                     decoder_nbrs=[dec[j] for j in b])
        lo, _, w, mom = O.train_step(w, ocfg, batch, lr=2e-3, weight_decay=1e-4, state=mom)
        assert abs(float(lg[i]) - float(lo)) < (1e-5 if i == 0 else 2e-4) * abs(float(lo)), (i, float(lg[i]), float(lo))


def _cfg_of(model):
    from tests.test_configs_gpu import model_cfg
    return model_cfg(model)


def test_trainstep_vx_fresh_dicts_every_step():
    """the reference's own loader keeps the graphs on the host and uploads them per step (move_to_device, static_trainer.py:192-193): new dict
    objects, new tensors every step.  Their per-sample plans are built per step without a host synchronisation; the step still replays."""
    from gaot_amd.trainer import TrainStep
    nS, B, N = 6, 3, 2048
    model, sd, _ = make_model(3, 1, [32, 32], radius=0.066, seed=4)
    lat, xs, enc, dec = _dataset(nS, N, seed=13, lat_sizes=(32, 32), radius=0.066)
    g = torch.Generator().manual_seed(1)
    P_all, T_all = torch.randn(nS, N, 3, generator=g), torch.randn(nS, N, 1, generator=g)
    batches = _shuffled_batches(nS, B, 5, seed=2)
    latd, xd = lat.to(dev()), torch.stack(xs).to(dev())
    out = {}
    for fresh in (True, False):
        from gaot_amd.model.gaot import GAOT
        m = GAOT(3, 1, _cfg_of(model))
        m.load_state_dict(sd)
        m.to(dev()).train()
        ts = TrainStep(m, lr=2e-3, weight_decay=1e-4, use_graph=True)
        kept_e = [[csr_dict(c) for c in row] for row in enc]
        kept_d = [[csr_dict(c) for c in row] for row in dec]
        up = (lambda rows, kept, b: [[csr_dict(c) for c in rows[i]] for i in b]) if fresh else (lambda rows, kept, b: [kept[i] for i in b])
        b0 = batches[0]
        ts.bind(P_all[b0].to(dev()), T_all[b0].to(dev()), latent_tokens_coord=latd, xcoord=xd[b0], encoder_nbrs=up(enc, kept_e, b0), decoder_nbrs=up(dec, kept_d, b0))
        ls = [ts.step(P_all[b].to(dev()), T_all[b].to(dev()), xcoord=xd[b], encoder_nbrs=up(enc, kept_e, b), decoder_nbrs=up(dec, kept_d, b)).clone() for b in batches]
        torch.cuda.synchronize()
        out[fresh] = torch.stack(ls).cpu()
    assert torch.equal(out[True], out[False])


def test_auto_graph_vx_replays_shuffled_compositions_and_follows_coordinates():
    """autograph.py in vx mode (the reference's variable-coordinate loop, static_trainer.py:180-202 inside optimizers.py:247-257, unchanged):
    from its third step on every batch -- any composition of the resident per-sample graphs, coordinates as new tensors each step, also EDITED
    coordinates -- is served by the captured graphs and gives what the eager path gives."""
    from gaot_amd.model.gaot import GAOT
    nS, B, N, steps = 8, 4, 2048, 9
    model, sd, _ = make_model(3, 1, [32, 32], radius=0.066, seed=6)
    lat, xs, enc, dec = _dataset(nS, N, seed=17, lat_sizes=(32, 32), radius=0.066)
    # one bucket for the whole dataset would hide nothing here; the test is about compositions, so keep the batches in ONE bucket by
    # drawing permutations of the same 4 samples for the first steps and other samples later
    batches = [[0, 1, 2, 3], [3, 1, 0, 2], [2, 0, 3, 1], [1, 3, 2, 0], [0, 2, 1, 3], [3, 2, 1, 0], [0, 1, 2, 3], [2, 3, 0, 1], [1, 0, 3, 2]]
    g = torch.Generator().manual_seed(3)
    P_all, T_all = torch.randn(nS, N, 3, generator=g), torch.randn(nS, N, 1, generator=g)
    latd, xd = lat.to(dev()), torch.stack(xs).to(dev())
    encd = [[csr_dict(c) for c in row] for row in enc]
    decd = [[csr_dict(c) for c in row] for row in dec]
    runs = {}
    for auto in (True, False):
        m = GAOT(3, 1, _cfg_of(model))
        m.load_state_dict(sd)
        m.to(dev()).train()
        m.auto_graph = auto
        opt = torch.optim.AdamW(m.parameters(), lr=2e-3, weight_decay=1e-4)
        losses, which = [], []
        for i, b in enumerate(batches):
            xc = xd[b].clone()
            if i == steps - 1:
                xc = xc * 1.0001          # edited coordinates with the same graphs: the captured forward re-derives the geometry arrays
            opt.zero_grad()
            out = m(latent_tokens_coord=latd, pndata=P_all[b].to(dev()), xcoord=xc, encoder_nbrs=[encd[j] for j in b], decoder_nbrs=[decd[j] for j in b])
            which.append(type(out.grad_fn).__name__)
            loss = torch.nn.functional.mse_loss(out, T_all[b].to(dev()))
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        runs[auto] = (losses, torch.cat([q.detach().reshape(-1) for q in m.parameters()]).cpu(), sum(w == "_GraphedStepBackward" for w in which), which)
    la, wa, ra, which = runs[True]
    lb, wb, rb, _ = runs[False]
    # the first forward re-homes the fused weights (new storage = new key), the second is that key's first sight, from the third on: graphs
    assert rb == 0 and ra == steps - 2, (ra, which)
    assert max(abs(a - b) / abs(b) for a, b in zip(la, lb)) < 1e-5, (la, lb)
    assert float((wa - wb).abs().max()) < 2e-5


def test_a_captured_no_grad_vx_forward_keeps_its_unions_through_cache_eviction():
    """A hipGraph captured over an EVALUATION-mode vx forward (what a rollout runner does per step) reads the composed unions of
    plan.merged_geometry by raw address.  Those unions are pinned at capture: more than _MERGE_CACHE_MAX other compositions passing through the
    cache afterwards (validation batches, a shuffling loader) must neither evict nor release them -- the replay still gives the captured
    composition's result."""
    from gaot_amd import plan as P
    nS, B, N = 12, 3, 2048
    model, sd, _ = make_model(3, 1, [32, 32], radius=0.066, seed=8)
    lat, xs, enc, dec = _dataset(nS, N, seed=23, lat_sizes=(32, 32), radius=0.066)
    latd, xd = lat.to(dev()), torch.stack(xs).to(dev())
    encd = [[csr_dict(c) for c in row] for row in enc]
    decd = [[csr_dict(c) for c in row] for row in dec]
    model.to(dev()).eval()
    g = torch.Generator().manual_seed(5)
    p = torch.randn(B, N, 3, generator=g).to(dev())
    b0 = [0, 1, 2]
    x0 = xd[b0].contiguous()
    kw0 = dict(latent_tokens_coord=latd, xcoord=x0, encoder_nbrs=[encd[i] for i in b0], decoder_nbrs=[decd[i] for i in b0])
    with torch.no_grad():
        want = model(pndata=p, **kw0).clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            model(pndata=p, **kw0)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, capture_error_mode="thread_local"):
            y = model(pndata=p, **kw0)
        pinned = [v[0] for v in P._MERGE_CACHE.values() if getattr(v[0], "pinned", False)]
        assert len(pinned) == 2                      # the encoder's and the decoder's union of the captured composition
        # many other compositions pass through the cache
        for k in range(P._MERGE_CACHE_MAX + 3):
            b = [(k + 3) % nS, (k + 5) % nS, (k + 8) % nS]
            model(pndata=p, latent_tokens_coord=latd, xcoord=xd[b].contiguous(), encoder_nbrs=[encd[i] for i in b], decoder_nbrs=[decd[i] for i in b])
        assert all(any(v[0] is u for v in P._MERGE_CACHE.values()) for u in pinned)
        assert all(u.plan._coord_cache for u in pinned)          # not released
        p2 = p * 1.0
        gr.replay()
        torch.cuda.synchronize()
    assert torch.equal(y, want)


def test_vx_static_path_3d_clouds_equal_composed_path_and_oracle():
    """the same check on 3-D point clouds (coord_dim 3: three-coordinate rows in the compose kernel, the 3 x 3 covariance eigenvalues per row)"""
    from gaot_amd import plan as P
    from gaot_amd import ops
    from oracle import gaot_oracle as O
    from tests._golden import fp32_noise, unfloored_ratio
    from tests._workloads import shell_points
    B, N = 3, 3000
    model, sd, ocfg = make_model(3, 1, [8, 8, 8], d=3, C=48, hidden=192, heads=4, radius=0.3, seed=12)
    g = torch.Generator().manual_seed(12)
    lat = grid([8, 8, 8])
    xs = [shell_points(N, g) for _ in range(B)]
    enc = [[O.radius_csr(x, lat, 0.3)] for x in xs]
    dec = [[O.radius_csr(lat, x, 0.3)] for x in xs]
    p, tgt = torch.randn(B, N, 3, generator=g), torch.randn(B, N, 1, generator=g)
    x = torch.stack(xs)
    model.to(dev()).train()
    fk = dict(latent_tokens_coord=lat.to(dev()), xcoord=x.to(dev()),
              encoder_nbrs=[[csr_dict(c) for c in row] for row in enc], decoder_nbrs=[[csr_dict(c) for c in row] for row in dec])
    res = {}
    for static in (True, False):
        P.VX_STATIC = static
        try:
            model.zero_grad(set_to_none=True)
            pred = model(pndata=p.to(dev()), **fk)
            loss = ops.mse_loss(pred, tgt.to(dev()))
            loss.backward()
            torch.cuda.synchronize()
            res[static] = (pred.detach().cpu(), float(loss.detach()), {k: q.grad.detach().cpu().clone() for k, q in model.named_parameters() if q.grad is not None})
        finally:
            P.VX_STATIC = True
    (ya, la, ga), (yb, lb, gb) = res[True], res[False]
    assert rel_l2(ya, yb) < 2e-6 and abs(la - lb) < 1e-6 * abs(lb)
    batch = dict(latent=lat, xcoord=x, pndata=p, target=tgt, encoder_nbrs=enc, decoder_nbrs=dec)
    lo, go, _, _, po = O.train_step(sd, ocfg, batch, return_pred=True)
    assert rel_l2(ya, po) < OUT_TOL and abs(la - float(lo)) < LOSS_TOL * abs(float(lo))
    ratio = unfloored_ratio(ga, {k: go[k] for k in ga}, fp32_noise(sd, ocfg, batch, go), GRAD_TOL)
    assert max(ratio.values()) <= 1.0, max(ratio, key=ratio.get)


@pytest.mark.parametrize("vx", [True, False])
def test_auto_graph_keeps_replaying_with_neighbour_sub_sampling(vx):
    """The unchanged reference loop on a model with `sampling_strategy='ratio'`: autograph captures the step WITH the device-side draw inside
    (plan.DropPlan), replays it from the third sight on, and every replay trains on another subset -- the kept edge count moves from step to
    step, the loss stays finite and close to the full-graph model's on the same data.  vx: caller-supplied per-sample lists (static unions);
    fx: module-owned lists."""
    from gaot_amd.model.gaot import GAOT
    from gaot_amd import plan as P
    B, N = 3, 2048
    model, sd, _ = make_model(3, 1, [32, 32], radius=0.066, seed=14, precompute=vx, sampling_strategy="ratio", sample_ratio=0.7)
    lat, xs, enc, dec = _dataset(B, N, seed=19, lat_sizes=(32, 32), radius=0.066)
    latd = lat.to(dev())
    xd = torch.stack(xs).to(dev()) if vx else xs[0].to(dev())
    kw = dict(encoder_nbrs=[[csr_dict(c) for c in row] for row in enc], decoder_nbrs=[[csr_dict(c) for c in row] for row in dec]) if vx else {}
    g = torch.Generator().manual_seed(4)
    data = [(torch.randn(B, N, 3, generator=g).to(dev()), torch.randn(B, N, 1, generator=g).to(dev())) for _ in range(7)]
    m = GAOT(3, 1, _cfg_of(model))
    m.load_state_dict(sd)
    m.to(dev()).train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-4)
    which, losses, kept = [], [], []
    for p_, t_ in data:
        opt.zero_grad()
        out = m(latent_tokens_coord=latd, pndata=p_, xcoord=xd.clone(), **kw)
        which.append(type(out.grad_fn).__name__)
        loss = torch.nn.functional.mse_loss(out, t_)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        if vx:
            base = next(iter(m.encoder._static_unions.values())).plan
        else:
            base = P.plan_for(next(iter(m.encoder.neighbor_cache.values()))[0], N)
        kept.append(int(next(iter(base._drops.values())).e_dev.item()))
    replays = sum(w == "_GraphedStepBackward" for w in which)
    assert replays == len(data) - 2, which
    assert all(l == l and l < 10 for l in losses), losses
    assert len(set(kept[2:])) >= 3, kept                       # every replay draws anew
    full = sum(int(e[0][0].numel()) for e in enc) if vx else int(next(iter(m.encoder.neighbor_cache.values()))[0]["neighbors_index"].numel())
    assert all(abs(k / full - 0.7) < 0.03 for k in kept), (kept, full)


def test_trainstep_vx_multiscale_shuffled_replay_equals_eager():
    """two scales with learned scale weights: four static unions per batch (encoder and decoder x two radii), the decoder's scale weights read
    the FIRST sample's coordinates of every batch (magno.py:610-612) from the step's static coordinate buffer -- replay == eager bit for bit
    over shuffled compositions, staged backward (the data-parallel schedule) included."""
    from gaot_amd.model.gaot import GAOT
    from gaot_amd.trainer import TrainStep
    from oracle import gaot_oracle as O
    nS, B, N = 6, 3, 2048
    model, sd, _ = make_model(3, 1, [32, 32], radius=0.066, seed=15, scales=[1.0, 0.5], use_scale_weights=True)
    g = torch.Generator().manual_seed(15)
    lat = grid([32, 32])
    xs = [naca_points(N, g, 0.15) for _ in range(nS)]
    enc = [[O.radius_csr(x, lat, 0.066 * s) for s in (1.0, 0.5)] for x in xs]
    dec = [[O.radius_csr(lat, x, 0.066 * s) for s in (1.0, 0.5)] for x in xs]
    P_all, T_all = torch.randn(nS, N, 3, generator=g), torch.randn(nS, N, 1, generator=g)
    batches = _shuffled_batches(nS, B, 5, seed=8)
    latd, xd = lat.to(dev()), torch.stack(xs).to(dev())
    encd = [[csr_dict(c) for c in row] for row in enc]
    decd = [[csr_dict(c) for c in row] for row in dec]
    runs = {}
    for graph, staged in ((True, False), (False, False), (True, True)):
        m = GAOT(3, 1, _cfg_of(model))
        m.load_state_dict(sd)
        m.to(dev()).train()
        ts = TrainStep(m, lr=2e-3, weight_decay=1e-4, use_graph=graph, staged=staged)
        b0 = batches[0]
        ts.bind(P_all[b0].to(dev()), T_all[b0].to(dev()), latent_tokens_coord=latd, xcoord=xd[b0], encoder_nbrs=[encd[i] for i in b0],
                decoder_nbrs=[decd[i] for i in b0])
        ls = [ts.step(P_all[b].to(dev()), T_all[b].to(dev()), xcoord=xd[b], encoder_nbrs=[encd[i] for i in b], decoder_nbrs=[decd[i] for i in b]).clone()
              for b in batches]
        torch.cuda.synchronize()
        assert len(m.encoder._static_unions) >= 2 and len(m.decoder._static_unions) >= 2
        runs[(graph, staged)] = (torch.stack(ls).cpu(), torch.cat([q.detach().reshape(-1) for q in m.parameters()]).cpu())
    assert torch.equal(runs[(True, False)][0], runs[(False, False)][0]) and torch.equal(runs[(True, False)][1], runs[(False, False)][1])
    # the staged schedule groups the weight-gradient launches differently: fp32 rounding, not bits
    assert float((runs[(True, True)][1] - runs[(True, False)][1]).abs().max()) < 2e-5
    assert float((runs[(True, True)][0] - runs[(True, False)][0]).abs().max()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["vx", "fx"])
def test_trainstep_multiscale_gradients_equal_plain_autograd(mode):
    """the scales of a multiscale MAGNO share every weight: the second use of a parameter in a pass takes autograd's ordinary accumulation and
    the first use must then write its registered slice at once, column blocks handed out by split_cols included (a strided block outside the
    grouped launch).  The flat buffer after one TrainStep pass (eager and captured) == the gradients of a plain loss.backward()."""
    from gaot_amd.model.gaot import GAOT
    from gaot_amd.trainer import TrainStep
    from gaot_amd import ops
    from oracle import gaot_oracle as O
    B, N = 2, 2048
    model, sd, _ = make_model(3, 1, [32, 32], radius=0.066, seed=16, scales=[1.0, 0.5], use_scale_weights=True)
    g = torch.Generator().manual_seed(16)
    lat = grid([32, 32])
    xs = [naca_points(N, g, 0.15) for _ in range(B)]
    p, t = torch.randn(B, N, 3, generator=g).to(dev()), torch.randn(B, N, 1, generator=g).to(dev())
    if mode == "vx":
        kw = dict(latent_tokens_coord=lat.to(dev()), xcoord=torch.stack(xs).to(dev()),
                  encoder_nbrs=[[csr_dict(O.radius_csr(x, lat, 0.066 * s)) for s in (1.0, 0.5)] for x in xs],
                  decoder_nbrs=[[csr_dict(O.radius_csr(lat, x, 0.066 * s)) for s in (1.0, 0.5)] for x in xs])
    else:
        kw = dict(latent_tokens_coord=lat.to(dev()), xcoord=xs[0].to(dev()),
                  encoder_nbrs=[csr_dict(O.radius_csr(xs[0], lat, 0.066 * s)) for s in (1.0, 0.5)],
                  decoder_nbrs=[csr_dict(O.radius_csr(lat, xs[0], 0.066 * s)) for s in (1.0, 0.5)])
    ref = GAOT(3, 1, _cfg_of(model))
    ref.load_state_dict(sd)
    ref.to(dev()).train()
    loss = ops.mse_loss(ref(pndata=p, **kw), t)
    loss.backward()
    want = {k: q.grad.clone() for k, q in ref.named_parameters()}
    del loss
    ref.zero_grad(set_to_none=True)
    for graph in (False, True):
        m = GAOT(3, 1, _cfg_of(model))
        m.load_state_dict(sd)
        m.to(dev()).train()
        ts = TrainStep(m, lr=0.0, weight_decay=0.0, use_graph=graph)
        ts.bind(p, t, **kw)
        for _ in range(2):          # (the second pass finds the buffer as the first left it: nothing may carry over)
            ts.step()
        torch.cuda.synchronize()
        for k, q in m.named_parameters():
            err = float((q.grad - want[k]).norm()) / max(float(want[k].norm()), 1e-30)
            assert err < 2e-5, (mode, graph, k, err)
