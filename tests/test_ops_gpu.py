"""-m gpu: every HIP kernel behind the C ABI against a float64 / oracle computation of the same op."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def maxrel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 70, 45), (8192, 256, 256), (1000, 1, 64), (77, 33, 7),
                                   (2048, 64, 4), (513, 257, 130), (64, 1024, 256)])
def test_gemm_nt_nn_tn(M, N, K):
    from gaot_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g)
    gy = torch.randn(M, N, generator=g)
    xd, wd, gd = x.to(dev()), w.to(dev()), gy.to(dev())
    y = ops.linear_nt(xd, wd)
    assert rel(y, x.double() @ w.double().t()) < 2e-6
    dx = ops.matmul_nn(gd, wd)
    assert rel(dx, gy.double() @ w.double()) < 2e-6
    dw = ops.matmul_tn(gd, xd)
    assert rel(dw, gy.double().t() @ x.double()) < 2e-6


def test_gemm_epilogues():
    from gaot_amd import ops, _lib as L
    g = torch.Generator().manual_seed(5)
    M, N, K, P = 384, 96, 64, 128
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / 8
    b, rb, rs, res = torch.randn(N, generator=g), torch.randn(P, N, generator=g), torch.rand(M, generator=g) + 0.5, torch.randn(M, N, generator=g)
    X, W, Bv, RB, RS, RES = [t.to(dev()) for t in (x, w, b, rb, rs, res)]
    z_ref = ((x.double() @ w.double().t() + b.double() + rb.double().repeat(M // P, 1)) * rs.double()[:, None])
    aux = torch.empty(M, N, device=dev())
    y = ops.linear_nt(X, W, bias=Bv, rowbias=RB, rowbias_period=P, ld_rowbias=N, rowscale=RS, act=L.ACT_GELU, aux_out=aux,
                      ld_aux=N, residual=RES, ldr=N)
    assert rel(aux, z_ref) < 2e-6
    assert rel(y, torch.nn.functional.gelu(z_ref) + res.double()) < 2e-6
    y2 = ops.linear_nt(X, W, bias=Bv, act=L.ACT_RELU)
    assert rel(y2, torch.relu(x.double() @ w.double().t() + b.double())) < 2e-6
    # backward activations fused into the input-gradient product
    zz = torch.randn(M, K, generator=g)
    gy = torch.randn(M, N, generator=g)
    zt = zz.double().requires_grad_(True)
    torch.nn.functional.gelu(zt).backward(gy.double() @ w.double())
    d = ops.matmul_nn(gy.to(dev()), W, act=L.ACT_GELU_BWD, aux_in=zz.to(dev()), ld_aux=K)
    assert rel(d, zt.grad) < 3e-6
    d2 = ops.matmul_nn(gy.to(dev()), W, act=L.ACT_RELU_BWD, aux_in=zz.to(dev()), ld_aux=K)
    assert rel(d2, (gy.double() @ w.double()) * (zz.double() > 0)) < 2e-6
    # split input (cat([x, x2]) @ W^T) and explicit split-K
    x2 = torch.randn(M, 32, generator=g)
    w2 = torch.randn(N, K + 32, generator=g)
    y3 = torch.empty(M, N, device=dev())
    ops.gemm(M, N, K + 32, X, K, 1, w2.to(dev()), K + 32, 1, y3, N, A2=x2.to(dev()), lda2=32, k_split=K)
    assert rel(y3, torch.cat([x, x2], 1).double() @ w2.double().t()) < 2e-6
    y4 = torch.empty(M, N, device=dev())
    ops.gemm(M, N, K, X, K, 1, W, K, 1, y4, N, bias=Bv, split_k=2)
    assert rel(y4, x.double() @ w.double().t() + b.double()) < 2e-6


def test_gemm_strided_views():
    from gaot_amd import ops
    g = torch.Generator().manual_seed(9)
    big = torch.randn(200, 160, generator=g).to(dev())
    w = torch.randn(48, 128, generator=g).to(dev())
    y = ops.linear_nt(big[:, 32:96], w[:, 64:])            # lda = 160, ldb = 128, offsets
    assert rel(y, big[:, 32:96].double().cpu() @ w[:, 64:].double().cpu().t()) < 2e-6


def test_linear_autograd_matches_torch():
    from gaot_amd import ops
    g = torch.Generator().manual_seed(11)
    B, S, K, K2, N = 3, 64, 32, 32, 40
    x, x2 = torch.randn(B, S, K, generator=g), torch.randn(B, S, K2, generator=g)
    w, b = torch.randn(N, K + K2, generator=g) / 6, torch.randn(N, generator=g)
    res, rb = torch.randn(B, S, N, generator=g), torch.randn(S, N, generator=g)
    leaves = [t.clone().double().requires_grad_(True) for t in (x, w, b, res, rb, x2)]
    ref = torch.cat([leaves[0], leaves[5]], -1) @ leaves[1].t() + leaves[2] + leaves[3] + leaves[4][None]
    go = torch.randn(B, S, N, generator=g)
    ref.backward(go.double())
    dl = [t.clone().to(dev()).requires_grad_(True) for t in (x, w, b, res, rb, x2)]
    out = ops.linear(dl[0], dl[1], dl[2], residual=dl[3], rowbias=dl[4], x2=dl[5])
    out.backward(go.to(dev()))
    assert rel(out, ref) < 2e-6
    for a, r in zip(dl, leaves):
        assert rel(a.grad, r.grad) < 3e-6


def test_mlp_chain_autograd():
    from gaot_amd import ops
    g = torch.Generator().manual_seed(13)
    E = 777
    x = torch.randn(E, 4, generator=g)
    ws = [torch.randn(16, 4, generator=g) / 2, torch.randn(16, 16, generator=g) / 4, torch.randn(8, 16, generator=g) / 4]
    bs = [torch.randn(16, generator=g) * 0.1, torch.randn(16, generator=g) * 0.1, torch.randn(8, generator=g) * 0.1]
    for acts in (["gelu", "gelu", "none"], ["relu", "relu", "relu"]):
        lw = [t.clone().double().requires_grad_(True) for t in ws]
        lb = [t.clone().double().requires_grad_(True) for t in bs]
        lx = x.clone().double().requires_grad_(True)
        h = lx
        for w_, b_, a in zip(lw, lb, acts):
            h = h @ w_.t() + b_
            h = torch.nn.functional.gelu(h) if a == "gelu" else (torch.relu(h) if a == "relu" else h)
        go = torch.randn(E, 8, generator=g)
        h.backward(go.double())
        dw = [t.clone().to(dev()).requires_grad_(True) for t in ws]
        db = [t.clone().to(dev()).requires_grad_(True) for t in bs]
        dx = x.clone().to(dev()).requires_grad_(True)
        out = ops.mlp_chain(dx, dw, db, acts)
        out.backward(go.to(dev()))
        assert rel(out, h) < 3e-6
        assert rel(dx.grad, lx.grad) < 1e-5
        for a_, r_ in zip(dw + db, lw + lb):
            assert rel(a_.grad, r_.grad) < 1e-5


# ------------------------------------------------------------------ geometry plan + GNO
def _random_geometry(seed, n, lat_sizes, radius, d=2):
    from oracle import gaot_oracle as O
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, d, generator=g) * 2 - 1
    lat = O.latent_grid(lat_sizes)
    return x, lat, O.radius_csr(x, lat, radius), O.radius_csr(lat, x, radius)


def _dict(csr):
    return {"neighbors_index": csr[0].to(dev()), "neighbors_row_splits": csr[1].to(dev())}


@pytest.mark.parametrize("d,n,lat,r", [(2, 500, [16, 16], 0.15), (2, 40, [16, 16], 0.1), (3, 400, [6, 6, 6], 0.5)])
def test_plan_arrays_attention_stats(d, n, lat, r):
    from gaot_amd.plan import plan_for
    from oracle import gaot_oracle as O
    x, latc, enc, _ = _random_geometry(3, n, lat, r, d)
    idx, sp = enc
    plan = plan_for(_dict(enc), n)
    assert torch.equal(plan.index[:plan.E].cpu().long(), idx) and torch.equal(plan.splits.cpu().long(), sp)
    qid, deg = O.edge_query_ids(sp)
    assert torch.equal(plan.edge_query[:plan.E].cpu().long(), qid)
    # transposed CSR: for every source j the ascending list of edges pointing at it
    tsp, te = plan.t_splits.cpu().long(), plan.t_edge[:plan.E].cpu().long()
    order = torch.argsort(idx * (plan.E + 1) + torch.arange(plan.E))
    assert torch.equal(te, order)
    assert torch.equal(tsp, torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(torch.bincount(idx, minlength=n), 0)]))
    X, Lc = x.to(dev()), latc.to(dev())
    # cosine attention
    xi, yj = latc[qid], x[idx]
    s = ((xi / xi.norm(dim=-1, keepdim=True).clamp_min(1e-12)) * (yj / yj.norm(dim=-1, keepdim=True).clamp_min(1e-12))).sum(-1)
    att = O.segment_softmax(s, qid, sp.numel() - 1)
    assert maxrel(plan.cosine_attention(X, Lc)[:plan.E], att) < 5e-6
    assert torch.equal(plan.edge_features(X, Lc).cpu(), torch.cat([yj, xi], -1))
    # geometry statistics (standardised): fp64 on device vs the oracle's fp32 LAPACK path
    st = plan.geo_stats(X, Lc).cpu()
    ref = O.geo_stats(x, latc, enc)
    assert (st - ref).abs().max() < 2e-4, (st - ref).abs().max()
    assert rel(st, ref) < 2e-5


def test_invalid_csr_is_rejected():
    from gaot_amd.plan import GeometryPlan
    idx = torch.tensor([0, 5, 1], device=dev())
    sp = torch.tensor([0, 2, 3], device=dev())
    with pytest.raises(ValueError):
        GeometryPlan(idx, sp, n_src=3)
    with pytest.raises(ValueError):
        GeometryPlan(torch.tensor([0, 1, 1], device=dev()), torch.tensor([0, 2, 1], device=dev()), n_src=3)


@pytest.mark.parametrize("C,B", [(64, 8), (16, 3), (6, 2), (10, 1)])
def test_gno_transform_fwd_bwd(C, B):
    from gaot_amd import ops
    from gaot_amd.plan import plan_for
    from oracle import gaot_oracle as O
    x, latc, enc, _ = _random_geometry(7, 600, [16, 16], 0.13)
    idx, sp = enc
    Q = sp.numel() - 1
    qid, _ = O.edge_query_ids(sp)
    g = torch.Generator().manual_seed(C)
    k = torch.randn(idx.numel(), C, generator=g)
    f = torch.randn(B, 600, C, generator=g)
    a = torch.rand(idx.numel(), generator=g)
    go = torch.randn(B, Q, C, generator=g)
    kr, fr = k.clone().double().requires_grad_(True), f.clone().double().requires_grad_(True)
    ref = O.seg_sum(kr[None] * fr[:, idx, :] * a.double()[None, :, None], qid, Q)
    ref.backward(go.double())
    plan = plan_for(_dict(enc), 600)
    kd, fd = k.to(dev()).requires_grad_(True), f.to(dev()).requires_grad_(True)
    out = ops.gno_transform(kd, fd, plan, a.to(dev()))
    out.backward(go.to(dev()))
    assert rel(out, ref) < 2e-6
    assert rel(kd.grad, kr.grad) < 3e-6
    assert rel(fd.grad, fr.grad) < 3e-6
    assert float(out[:, (sp[1:] - sp[:-1]) == 0].abs().max() if ((sp[1:] - sp[:-1]) == 0).any() else 0.0) == 0.0


def test_segment_softmax_and_sum():
    from gaot_amd import ops
    from gaot_amd.plan import plan_for
    from oracle import gaot_oracle as O
    x, latc, enc, _ = _random_geometry(8, 300, [16, 16], 0.16)
    idx, sp = enc
    Q = sp.numel() - 1
    qid, deg = O.edge_query_ids(sp)
    g = torch.Generator().manual_seed(1)
    s = torch.randn(idx.numel(), generator=g)
    gy = torch.randn(idx.numel(), generator=g)
    sr = s.clone().double().requires_grad_(True)
    O.segment_softmax(sr, qid, Q).backward(gy.double())
    plan = plan_for(_dict(enc), 300)
    sd_ = s.to(dev()).requires_grad_(True)
    a = ops.segment_softmax(sd_, plan)
    a.backward(gy.to(dev()))
    assert rel(a, O.segment_softmax(s.double(), qid, Q)) < 2e-6
    assert rel(sd_.grad, sr.grad) < 1e-5
    xx = torch.randn(2, idx.numel(), 12, generator=g)
    inv = 1.0 / deg.clamp(min=1).float()
    out = ops.segment_sum(xx.to(dev()), plan, inv.to(dev()))
    assert rel(out, O.seg_sum(xx.double(), qid, Q) * inv.double()[None, :, None]) < 2e-6


# ------------------------------------------------------------------ processor kernels
@pytest.mark.parametrize("M,D", [(1000, 256), (37, 48), (8192, 64), (4099, 384), (513, 512)])
def test_rmsnorm(M, D):
    from gaot_amd import ops
    from oracle import gaot_oracle as O
    g = torch.Generator().manual_seed(M + D)
    x, w, go = torch.randn(M, D, generator=g), torch.rand(D, generator=g) + 0.5, torch.randn(M, D, generator=g)
    xr, wr = x.clone().double().requires_grad_(True), w.clone().double().requires_grad_(True)
    O.rms_norm(xr, wr, 1e-6).backward(go.double())
    xd, wd = x.to(dev()).requires_grad_(True), w.to(dev()).requires_grad_(True)
    y = ops.rms_norm(xd, wd, 1e-6)
    y.backward(go.to(dev()))
    assert rel(y, O.rms_norm(x.double(), w.double(), 1e-6)) < 2e-6
    assert rel(xd.grad, xr.grad) < 3e-6 and rel(wd.grad, wr.grad) < 3e-6


def test_swiglu():
    from gaot_amd import ops
    g = torch.Generator().manual_seed(2)
    u, go = torch.randn(5, 64, 128, generator=g), torch.randn(5, 64, 64, generator=g)
    ur = u.clone().double().requires_grad_(True)
    ref = torch.nn.functional.silu(ur[..., :64]) * ur[..., 64:]
    ref.backward(go.double())
    ud = u.to(dev()).requires_grad_(True)
    out = ops.swiglu(ud)
    out.backward(go.to(dev()))
    assert rel(out, ref) < 2e-6 and rel(ud.grad, ur.grad) < 3e-6


@pytest.mark.parametrize("B,S,H,Hkv,D", [(2, 256, 4, 4, 32), (1, 1024, 8, 8, 32), (2, 64, 4, 4, 8), (2, 200, 4, 4, 12),
                                         (1, 130, 2, 2, 64), (2, 96, 4, 2, 16), (1, 33, 1, 1, 32), (2, 203, 4, 2, 32),
                                         (1, 129, 2, 1, 32), (2, 128, 4, 2, 32), (1, 64, 2, 2, 32),
                                         (2, 200, 4, 2, 48), (1, 256, 2, 2, 64), (1, 512, 4, 4, 48), (1, 70, 2, 1, 36),
                                         (1, 160, 2, 2, 128), (2, 97, 2, 1, 96), (1, 300, 1, 1, 72)])
def test_attention_fwd_bwd(B, S, H, Hkv, D):
    from gaot_amd import ops, _lib
    g = torch.Generator().manual_seed(S + D)
    W = (H + 2 * Hkv) * D
    qkv = torch.randn(B, S, W, generator=g)
    go = torch.randn(B, S, H * D, generator=g)
    r = qkv.clone().double().requires_grad_(True)
    q = r[..., :H * D].reshape(B, S, H, D).transpose(1, 2)
    k = r[..., H * D:(H + Hkv) * D].reshape(B, S, Hkv, D).transpose(1, 2).repeat_interleave(H // Hkv, dim=1)
    v = r[..., (H + Hkv) * D:].reshape(B, S, Hkv, D).transpose(1, 2).repeat_interleave(H // Hkv, dim=1)
    p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), dim=-1)
    ref = (p @ v).transpose(1, 2).reshape(B, S, H * D)
    ref.backward(go.double())
    d = qkv.to(dev()).requires_grad_(True)
    out = ops.attention(d, H, Hkv, D)
    out.backward(go.to(dev()))
    assert rel(out, ref) < 3e-6
    assert rel(d.grad, r.grad) < 1e-5
    if 32 <= D <= 64 and D % 4 == 0:        # the default above is the split-bf16 pair of kernels (head_dim 32, and 36..64 in steps of 4); the fp32-MFMA kernels stay covered too
        old = _lib.load().gaot_debug_set_attention_split(0)
        try:
            d2 = qkv.to(dev()).requires_grad_(True)
            out2 = ops.attention(d2, H, Hkv, D)
            out2.backward(go.to(dev()))
        finally:
            _lib.load().gaot_debug_set_attention_split(old)
        assert rel(out2, ref) < 3e-6 and rel(d2.grad, r.grad) < 1e-5
        # the default (exact three-way splits of every operand) is as accurate as the fp32 MFMA
        assert rel(out, ref) < 2 * rel(out2, ref) + 1e-7 and rel(d.grad, r.grad) < 2 * rel(d2.grad, r.grad) + 1e-7
        # the 256-query / 256-key workgroup variants (picked by the heuristic only when they fill the chip) forced on
        old = _lib.load().gaot_debug_set_attention_split(2)
        try:
            d3 = qkv.to(dev()).requires_grad_(True)
            out3 = ops.attention(d3, H, Hkv, D)
            out3.backward(go.to(dev()))
        finally:
            _lib.load().gaot_debug_set_attention_split(old)
        assert rel(out3, ref) < 3e-6 and rel(d3.grad, r.grad) < 1e-5
        if S % 64 == 0 and D == 32:     # mode 2 ran the plain 8-wave forward; the software-pipelined variant stays covered too
            lib = _lib.load()
            old, oldp = lib.gaot_debug_set_attention_split(2), lib.gaot_debug_set_attention_pipe(1)
            try:
                with torch.no_grad():
                    out4 = ops.attention(qkv.to(dev()), H, Hkv, D)
            finally:
                lib.gaot_debug_set_attention_split(old); lib.gaot_debug_set_attention_pipe(oldp)
            assert rel(out4, ref) < 3e-6 and rel(out4, out3) < 3e-6      # (the pipelined variant splits P three ways, the default two)


def _dropout_factor(seed: int, B, H, S, p):
    """host restatement of attention.hip attn_drop_factor: splitmix64 finaliser of seed + ((b*H + h)*S + q)*S + k"""
    import numpy as np
    idx = np.arange(B * H * S * S, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = np.uint64(seed & (2 ** 64 - 1)) + idx
        x ^= x >> np.uint64(30); x *= np.uint64(0xbf58476d1ce4e5b9)
        x ^= x >> np.uint64(27); x *= np.uint64(0x94d049bb133111eb)
        x ^= x >> np.uint64(31)
    keep = (x >> np.uint64(32)) < np.uint64(min(int((1.0 - p) * 2 ** 32), 2 ** 32 - 1))
    return torch.from_numpy(keep.reshape(B, H, S, S).astype(np.float64) / (1.0 - p))


@pytest.mark.parametrize("B,S,H,Hkv,D,p", [(2, 96, 4, 4, 32, 0.1), (1, 130, 2, 1, 64, 0.25), (2, 70, 4, 2, 16, 0.5), (1, 150, 2, 2, 96, 0.2), (1, 64, 1, 1, 128, 0.3)])
def test_attention_dropout_matches_float64_with_the_same_mask(B, S, H, Hkv, D, p):
    """attn.py:110-114: softmax, then dropout of the attention weights, then the product with V.  The mask cannot equal torch's
    (another generator); the kernel's own mask is rebuilt on the host from the seed word and the math is checked against float64"""
    from gaot_amd import ops
    g = torch.Generator().manual_seed(S + D)
    W = (H + 2 * Hkv) * D
    qkv = torch.randn(B, S, W, generator=g)
    go = torch.randn(B, S, H * D, generator=g)
    ops.seed_dropout(1234, dev())
    d = qkv.to(dev()).requires_grad_(True)
    out = ops.attention(d, H, Hkv, D, dropout_p=p)
    out.backward(go.to(dev()))
    seed = int(ops._LAST_DROPOUT_SEED[0].cpu().item())
    fac = _dropout_factor(seed, B, H, S, p)
    assert abs(float((fac > 0).double().mean()) - (1.0 - p)) < 0.02          # the keep rate
    r = qkv.clone().double().requires_grad_(True)
    q = r[..., :H * D].reshape(B, S, H, D).transpose(1, 2)
    k = r[..., H * D:(H + Hkv) * D].reshape(B, S, Hkv, D).transpose(1, 2).repeat_interleave(H // Hkv, dim=1)
    v = r[..., (H + Hkv) * D:].reshape(B, S, Hkv, D).transpose(1, 2).repeat_interleave(H // Hkv, dim=1)
    pr = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), dim=-1) * fac
    ref = (pr @ v).transpose(1, 2).reshape(B, S, H * D)
    ref.backward(go.double())
    assert rel(out, ref) < 3e-6
    assert rel(d.grad, r.grad) < 1e-5
    # a second call draws another mask (the counter lives on the device); re-seeding reproduces the first
    out2 = ops.attention(d.detach(), H, Hkv, D, dropout_p=p)
    assert rel(out2, ref) > 1e-2
    ops.seed_dropout(1234, dev())
    assert torch.equal(ops.attention(d.detach(), H, Hkv, D, dropout_p=p), out.detach())


def test_attention_dropout_draws_a_new_mask_at_every_graph_replay():
    from gaot_amd import ops
    qkv = torch.randn(1, 64, 3 * 2 * 32, device=dev())
    ops.seed_dropout(7, dev())
    ops.attention(qkv, 2, 2, 32, dropout_p=0.3)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        y = ops.attention(qkv, 2, 2, 32, dropout_p=0.3)
    gr.replay(); a = y.clone()
    gr.replay(); b = y.clone()
    assert not torch.equal(a, b) and torch.isfinite(a).all() and torch.isfinite(b).all()


def test_attention_peaked_softmax():
    """one key dominates one query (running-max rescale path) -- full-tensor check against fp64"""
    from gaot_amd import ops
    g = torch.Generator().manual_seed(4)
    B, S, H, D = 1, 256, 2, 32
    qkv = torch.randn(B, S, 3 * H * D, generator=g)
    qkv[0, 17, :D] *= 6.0
    qkv[0, 201, H * D:H * D + D] = qkv[0, 17, :D] * 1.5
    r = qkv.double()
    q = r[..., :H * D].reshape(B, S, H, D).transpose(1, 2)
    k = r[..., H * D:2 * H * D].reshape(B, S, H, D).transpose(1, 2)
    v = r[..., 2 * H * D:].reshape(B, S, H, D).transpose(1, 2)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), -1) @ v).transpose(1, 2).reshape(B, S, H * D)
    out = ops.attention(qkv.to(dev()), H, H, D)
    assert rel(out, ref) < 3e-6 and torch.isfinite(out).all()
    from gaot_amd import _lib
    lib = _lib.load()
    for pipe in (0, 1):      # the 8-wave forwards; pipelined: unconditional rescale, first tile from -inf
        old, oldp = lib.gaot_debug_set_attention_split(2), lib.gaot_debug_set_attention_pipe(pipe)
        try:
            out = ops.attention(qkv.to(dev()), H, H, D)
        finally:
            lib.gaot_debug_set_attention_split(old); lib.gaot_debug_set_attention_pipe(oldp)
        assert rel(out, ref) < 3e-6 and torch.isfinite(out).all()


@pytest.mark.parametrize("sizes,P,C", [([16, 16], 2, 8), ([8, 12], 4, 5), ([8, 8, 8], 2, 6), ([64, 64], 2, 64)])
def test_patchify_roundtrip(sizes, P, C):
    from gaot_amd import ops
    from oracle import gaot_oracle as O
    n = math.prod(sizes)
    x = torch.randn(2, n, C)
    t = ops.patchify(x.to(dev()), sizes, P)
    assert torch.equal(t.cpu(), O.patchify(x, sizes, P))
    assert torch.equal(ops.unpatchify(t, sizes, P).cpu(), x)


def test_reductions():
    from gaot_amd import ops
    for shape in [(1000, 72), (256, 256), (37, 4), (1000, 70)]:
        x = torch.randn(*shape)
        assert rel(ops.colsum(x.to(dev())), x.double().sum(0)) < 2e-6      # short matrix: single launch when N % 4 == 0
    x = torch.randn(5000, 70)
    assert rel(ops.colsum(x.to(dev())), x.double().sum(0)) < 2e-6          # two-stage path
    y = torch.randn(5, 33, 7)
    assert rel(ops.batchsum(y.to(dev()), 5), y.double().sum(0)) < 2e-6


def test_host_tensors_are_refused():
    from gaot_amd import ops
    with pytest.raises(RuntimeError):
        ops.linear(torch.randn(4, 4), torch.randn(4, 4))


@pytest.mark.parametrize("N,K,M", [(256, 256, 8192), (64, 64, 55592), (64, 1, 131072), (1, 64, 70000), (64, 7, 4096), (96, 40, 1000),
                                   (64, 9, 65536), (12, 48, 32768), (64, 16, 5000), (9, 64, 20000)])
def test_weight_grad_with_fused_bias_grad(N, K, M):
    """dW = dY^T X with the bias gradient (column sums of dY) produced by the same kernel (MFMA and skinny paths)"""
    from gaot_amd import ops
    g = torch.Generator().manual_seed(N + K)
    gy, x = torch.randn(M, N, generator=g), torch.randn(M, K, generator=g)
    db = torch.empty(N, device=dev())
    dw = ops.matmul_tn(gy.to(dev()), x.to(dev()), colsum_out=db)
    assert rel(dw, gy.double().t() @ x.double()) < 3e-6
    # column sums of N(0,1) data nearly cancel: measure the error against the summed magnitudes
    assert float((db.double().cpu() - gy.double().sum(0)).abs().max() / gy.double().abs().sum(0).max()) < 1e-6


@pytest.mark.parametrize("d,n,m,r", [(2, 16384, 4096, 0.033), (2, 3000, 5000, 0.05), (3, 20000, 4096, 0.12), (3, 500, 64, 0.9),
                                     (2, 50, 200, 1e-4)])
def test_hip_radius_search_equals_exact_search(d, n, m, r):
    from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch, _exact_pairwise
    g = torch.Generator().manual_seed(n + m)
    data = (torch.rand(n, d, generator=g) * 2 - 1).to(dev())
    q = (torch.rand(m, d, generator=g) * 2.6 - 1.3).to(dev())          # some queries outside the data's bounding box
    q[:5] = data[:5]                                                       # coincident points (distance exactly 0)
    got = NeighborSearch("auto")(data, q, r)
    ref = _exact_pairwise(data, q, torch.tensor(r, device=dev()), False)
    assert torch.equal(got["neighbors_row_splits"], ref["neighbors_row_splits"])
    assert torch.equal(got["neighbors_index"], ref["neighbors_index"])


def test_hip_radius_search_inclusive_boundary():
    from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch
    lattice = torch.stack(torch.meshgrid(torch.arange(5.), torch.arange(5.), indexing="ij"), -1).reshape(-1, 2).to(dev())
    qs = torch.tensor([[2., 2.], [0., 0.], [4., 1.], [10., 10.], [2.5, 2.5]], device=dev())
    out = NeighborSearch("native")(lattice, qs, 1.0)
    deg = (out["neighbors_row_splits"][1:] - out["neighbors_row_splits"][:-1]).tolist()
    assert deg == [5, 3, 4, 0, 4]                  # axis neighbours at distance exactly r are included
    assert out["neighbors_index"][:5].tolist() == [7, 11, 12, 13, 17]


@pytest.mark.parametrize("d,n,m,r,cap", [(2, 8192, 4096, 0.09, 32), (3, 20000, 4096, 0.2, 32), (2, 3000, 500, 0.3, 8)])
def test_hip_radius_search_torch_cluster_cap(d, n, m, r, cap):
    """the cell-list builder with the torch_cluster option: strict d^2 < r^2, the `cap` smallest data indices per query --
    against the host search with the same rule and (small slice) the literal restatement of the published kernel"""
    from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch, _exact_pairwise
    from oracle import gaot_oracle as O
    g = torch.Generator().manual_seed(n + m + cap)
    data = (torch.rand(n, d, generator=g) * 2 - 1)
    q = (torch.rand(m, d, generator=g) * 2.6 - 1.3)
    ns = NeighborSearch("torch_cluster") if cap == 32 else NeighborSearch("torch_cluster", max_num_neighbors=cap)
    got = ns(data.to(dev()), q.to(dev()), r)
    deg = got["neighbors_row_splits"][1:] - got["neighbors_row_splits"][:-1]
    assert int(deg.max()) == cap and int((deg == cap).sum()) > 10           # the cap binds on many rows
    ref = _exact_pairwise(data, q, torch.tensor(r), False, cap, True)
    # squared distances within rounding of r^2 may fall either side (fused vs separate multiply-add): compare as sets off the boundary
    if not (torch.equal(got["neighbors_row_splits"].cpu(), ref["neighbors_row_splits"]) and torch.equal(got["neighbors_index"].cpu(), ref["neighbors_index"])):
        gi, gs, ri, rs = got["neighbors_index"].cpu(), got["neighbors_row_splits"].cpu(), ref["neighbors_index"], ref["neighbors_row_splits"]
        bad = 0
        for i in range(m):
            a, b = set(gi[gs[i]:gs[i + 1]].tolist()), set(ri[rs[i]:rs[i + 1]].tolist())
            for j in a ^ b:
                bad += 1
                full = ((data - q[i]) ** 2).sum(-1)
                near = ((full - r * r).abs() < 1e-6 * r * r)
                assert bool(near.any()), (i, j)                                # a boundary point shifted the first-`cap` window
        assert bad < 20
    idx, sp = O.radius_csr_torch_cluster(data, q[:64], r, cap)
    assert torch.equal(got["neighbors_row_splits"][:65].cpu(), sp) and torch.equal(got["neighbors_index"][:int(sp[-1])].cpu(), idx)


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,F", [(8192, 256, 1024), (300, 64, 132), (77, 32, 20), (1024, 96, 256)])
def test_swiglu_ffn_fused_matches_unfused(M, K, F):
    """gate fused into the [w1;w3] GEMM epilogue / gate gradient fused into the dY w2 epilogue (attn.py:150-156)
    against the separate HIP kernels (same accumulators -> identical) and a float64 torch reference."""
    from gaot_amd import ops
    torch.manual_seed(M + K + F)
    dev = "cuda"
    x = torch.randn(M, K, device=dev, requires_grad=True)
    w1 = (torch.randn(F, K, device=dev) / K ** 0.5).requires_grad_()
    w3 = (torch.randn(F, K, device=dev) / K ** 0.5).requires_grad_()
    w2 = (torch.randn(K, F, device=dev) / F ** 0.5).requires_grad_()
    res = torch.randn(M, K, device=dev, requires_grad=True)
    dy = torch.randn(M, K, device=dev)
    assert ops._SwiGLUFFN.fusable(K, F)
    y = ops.swiglu_ffn(x, w1, w3, w2, residual=res)
    gf = torch.autograd.grad(y, [x, w1, w3, w2, res], dy)
    y0 = ops.linear(ops.swiglu(ops.linear(x, torch.cat([w1, w3], 0))), w2, residual=res)
    g0 = torch.autograd.grad(y0, [x, w1, w3, w2, res], dy)
    assert torch.equal(y, y0)
    for a, b in zip(gf, g0):
        assert rel(a, b) < 2e-6
    xd, w1d, w3d, w2d, rd = [t.detach().double().requires_grad_() for t in (x, w1, w3, w2, res)]
    yd = (torch.nn.functional.silu(xd @ w1d.T) * (xd @ w3d.T)) @ w2d.T + rd
    gd = torch.autograd.grad(yd, [xd, w1d, w3d, w2d, rd], dy.double())
    assert rel(y, yd) < 1e-5
    for a, b in zip(gf, gd):
        assert rel(a, b) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [5, 7, 11])
@pytest.mark.parametrize("M,N,K", [(200, 132, 64), (520, 260, 96), (1000, 64, 256), (4096, 512, 1024)])
def test_gemm_split_bf16_kernel(M, N, K, mode, request):
    """gemm_split.hip: fp32 operands split exactly into three bf16 pieces, six bf16 MFMAs per product.  Forced on
    (mode 5) for every layout / epilogue it serves; must be as accurate as the fp32 MFMA kernel (float64 reference),
    including operands whose rows differ by orders of magnitude.  mode 5 = 128x128 tiles, 7 = 64-row tiles, 11 = 256-row (8-wave)
    tiles wherever M >= 512.  Runs with ops.set_f32_pieces("bf16x3"): the three-piece kernels are what serves operands without magnitude
    words and the A/B of the default fp16-piece products (test_gemm_fp16_piece_products pins those, incl. their own dynamic-range bound)."""
    from gaot_amd import ops, _lib
    lib = _lib.load()
    request.addfinalizer(lambda old=ops.set_f32_pieces("bf16x3"): ops.set_f32_pieces(old))
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g) * torch.exp(3 * torch.randn(M, 1, generator=g))
    w, gy = torch.randn(N, K, generator=g), torch.randn(M, N, generator=g)
    b, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    Mk = M - M % 32
    xd, wd, gd = x.double(), w.double(), gy.double()
    z_ref = xd @ wd.t() + b.double()
    refs = dict(y=torch.nn.functional.gelu(z_ref) + res.double(), z=z_ref,
                dx=(gd @ wd), dxg=(gd @ wd) * 1.0, dw=gd[:Mk].t() @ xd[:Mk], db=gd[:Mk].sum(0))
    old = lib.gaot_debug_set_gemm_glds(mode)
    try:
        d = "cuda"
        z = torch.empty(M, N, device=d)
        y = ops.linear_nt(x.to(d), w.to(d), bias=b.to(d), act=_lib.ACT_GELU, aux_out=z, ld_aux=N, residual=res.to(d), ldr=N)
        assert lib.gaot_debug_last_gemm_path() == 3
        dx = ops.matmul_nn(gy.to(d), w.to(d))
        assert lib.gaot_debug_last_gemm_path() == (3 if N % 32 == 0 else 1)     # the reduction must be a multiple of 32
        dw, db = torch.empty(N, K, device=d), torch.empty(N, device=d)
        ops.gemm(N, K, Mk, gy.to(d), N, 0, x.to(d), K, 0, dw, K, split_k=3, colsum=db)
        assert lib.gaot_debug_last_gemm_path() == 3
        dw1 = torch.empty(N, K, device=d)
        ops.gemm(N, K, Mk, gy.to(d), N, 0, x.to(d), K, 0, dw1, K)          # no split-K: direct epilogue
        # mixed layout: A m-major, B k-major  (x^T stored [K, M] against w)
        xt = x.t().contiguous().to(d)
        y2 = torch.empty(M, N, device=d)
        ops.gemm(M, N, K, xt, M, 0, w.to(d), K, 1, y2, N) if M % 4 == 0 else None
    finally:
        lib.gaot_debug_set_gemm_glds(old)
    assert rel(y, refs["y"]) < 2e-6 and rel(z, refs["z"]) < 2e-6
    assert rel(dx, refs["dx"]) < 2e-6
    assert rel(dw, refs["dw"]) < 2e-6 and rel(dw1, refs["dw"]) < 2e-6
    assert float((db.double().cpu() - refs["db"]).abs().max() / gd[:Mk].abs().sum(0).max()) < 1e-6
    if M % 4 == 0:
        assert rel(y2, xd @ wd.t()) < 2e-6
    # element-wise: error relative to sum |a||b| stays at the fp32 rounding level (no lost low-order pieces)
    scale = xd.abs() @ wd.abs().t()
    assert float(((z.double().cpu() - z_ref).abs() / (scale + 1)).max()) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(8192, 2048, 256), (8192, 256, 1024), (4096, 768, 256), (520, 260, 96)])
def test_gemm_fp16_piece_products(M, N, K):
    """gaot_gemm_desc.pieces = 4, the way the default "f32" precision runs on the split tiles: every operand scaled by the power of two
    that puts its largest magnitude (a device word: published by its producer or computed by gaot_absmax_grouped) into [2^13, 2^14) and
    split into TWO fp16 pieces, both rounded to nearest (s x = h + m + e, |e| <= 2^-23 |s x|: at most the operand's last bit, zero for
    three values in four), three piece products on the f16 MFMA.
      * random normal operands at magnitudes from 1e-9 to 1e+9: all three product kinds and the grouped launch are AT LEAST as close to
        float64 as the three-piece bf16 products (fewer accumulation steps on the matrix pipe), < 6e-7;
      * the words follow the data: the same tensor objects refilled with 1e6 x larger values (in place, through torch) give the same
        relative error -- no stale scale, no overflow;
      * dynamic range INSIDE an operand: rows 1e-4 and rows 1e-7 below the largest keep fp32 accuracy (the latter through the per-row
        second pass of their tiles, round 5)."""
    from gaot_amd import ops, _lib
    lib = _lib.load()
    assert ops.precision() == "f32" and ops._F16_PIECES[0]
    g = torch.Generator().manual_seed(M + N + K)
    Mk = M - M % 32
    for mag_a, mag_b in ((1.0, 0.06), (1e-9, 30.0), (1e9, 1e-3)):
        x, w, dy = torch.randn(M, K, generator=g) * mag_a, torch.randn(N, K, generator=g) * mag_b, torch.randn(M, N, generator=g) * mag_a
        xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
        ref = {"nt": x.double() @ w.double().t(), "nn": dy.double() @ w.double(), "tn": dy[:Mk].double().t() @ x[:Mk].double()}
        run = {"nt": lambda: ops.linear_nt(xd, wd), "nn": lambda: ops.matmul_nn(dyd, wd), "tn": lambda: ops.matmul_tn(dyd[:Mk], xd[:Mk])}
        for kind in ("nt", "nn", "tn"):
            e4 = rel(run[kind](), ref[kind])
            split = lib.gaot_debug_last_gemm_path() == 3
            old = ops.set_f32_pieces("bf16x3")
            try:
                e3 = rel(run[kind](), ref[kind])
            finally:
                ops.set_f32_pieces(old)
            assert e4 < 6e-7 and e3 < 1e-6 and (not split or e4 < 1.05 * e3), (kind, mag_a, mag_b, e4, e3)
        if N % 4 == 0 and K % 4 == 0 and Mk >= 1024:
            out4, out3 = torch.empty(N, K, device="cuda"), torch.empty(N, K, device="cuda")
            ops.wgrad_launch([(dyd[:Mk], N, xd[:Mk], K, out4, K, None, N, K, Mk)])
            old = ops.set_f32_pieces("bf16x3")
            try:
                ops.wgrad_launch([(dyd[:Mk], N, xd[:Mk], K, out3, K, None, N, K, Mk)])
            finally:
                ops.set_f32_pieces(old)
            assert rel(out4, ref["tn"]) < 6e-7 and rel(out4, ref["tn"]) < 1.05 * rel(out3, ref["tn"])
            again = torch.empty_like(out4)
            ops.wgrad_launch([(dyd[:Mk], N, xd[:Mk], K, again, K, None, N, K, Mk)])
            assert torch.equal(again, out4)                      # deterministic
    # the words follow the data: refill the SAME tensor objects (torch ops bump their version: the cached words are dropped)
    e0 = rel(ops.linear_nt(xd, wd), ref["nt"])
    xd.mul_(2.0 ** 20); wd.mul_(2.0 ** -3)
    e1 = rel(ops.linear_nt(xd, wd), ref["nt"] * 2.0 ** 17)
    assert abs(e1 - e0) < 1e-9, (e0, e1)                          # power-of-two rescaling: bit-identical up to the scale
    # dynamic range inside an operand
    x = torch.randn(M, K, generator=g)
    x[1::4] *= 1e-4
    x[2::4] *= 1e-7
    y = ops.linear_nt(x.cuda(), w.cuda())
    if lib.gaot_debug_last_gemm_path() == 3:
        r = x.double() @ w.double().t()
        assert rel(y[0::4], r[0::4]) < 6e-7 and rel(y[1::4], r[1::4]) < 6e-7
        # round 5: rows more than 2^13 below the tensor's largest magnitude send their tile through the per-row second pass
        # (test_gemm_fp16_pieces_rows_of_any_magnitude): fp32 level there too (round 4 carried them to ~1e-4)
        assert rel(y[2::4], r[2::4]) < 6e-7


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(8192, 256, 256), (4096, 768, 256), (8192, 1024, 512), (520, 260, 96)])
def test_gemm_fp16_pieces_rows_of_any_magnitude(M, N, K):
    """The fp16 pieces scale an operand by ONE power of two per tensor; a row of the operand (a token of the activations, an output
    feature of the weights, a channel in the weight-gradient products: the index that survives into the output) more than 2^13 below
    the tensor's largest magnitude would lose relative precision.  Every workgroup tracks its operand rows' maxima while staging and
    recomputes such a tile with one power of two PER ROW (gemm_split.hip "Dynamic range of the fp16 pieces"); pre-split weight planes
    carry their verdict in the weight's magnitude word.  Here: one row / column of an operand 1e4 and 1e6 times larger (the "massive
    activation" pattern of trained transformers) or rows 1e-6 smaller -- every OTHER row / column of the result must stay at fp32 level
    against float64, in all three product kinds, the grouped weight-gradient launch and with pre-split planes; benign data must not
    take the second pass at all."""
    from gaot_amd import ops, _lib
    lib = _lib.load()
    assert ops.precision() == "f32" and ops._F16_PIECES[0]
    g = torch.Generator().manual_seed(M + 3 * N + K)
    Mk = M - M % 32
    x, w, dy = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.06, torch.randn(M, N, generator=g)
    TOL = 6e-7

    def others(t, dim, idx):
        keep = torch.ones(t.shape[dim], dtype=torch.bool)
        keep[idx] = False
        return t[keep] if dim == 0 else t[:, keep]

    def products(x, w, dy, planes):
        xd, dyd = x.cuda(), dy.cuda()
        wd = torch.nn.Parameter(w.cuda()) if planes else w.cuda()
        ops.begin_pass()
        if planes:
            ops.refresh_weight_amax([wd])
        lib.gaot_debug_split_redo_count(1)
        out = {}
        with torch.no_grad():
            out["nt"] = ops.linear_nt(xd, wd.detach() if planes else wd)
            split = lib.gaot_debug_last_gemm_path() == 3
            out["nn"] = ops.matmul_nn(dyd, wd.detach() if planes else wd)
            out["tn"] = ops.matmul_tn(dyd[:Mk], xd[:Mk])
            if N % 4 == 0 and K % 4 == 0 and Mk >= 1024:
                out["tng"] = torch.empty(N, K, device="cuda")
                ops.wgrad_launch([(dyd[:Mk], N, xd[:Mk], K, out["tng"], K, None, N, K, Mk)])
        torch.cuda.synchronize()
        return out, split, int(lib.gaot_debug_split_redo_count(1))

    def refs(x, w, dy):
        return {"nt": x.double() @ w.double().t(), "nn": dy.double() @ w.double(), "tn": dy[:Mk].double().t() @ x[:Mk].double()}

    # benign data: no tile takes the second pass
    for planes in (False, True):
        out, split, redone = products(x, w, dy, planes)
        r = refs(x, w, dy)
        assert redone == 0, (planes, redone)
        for k, v in out.items():
            assert rel(v, r["tn" if k == "tng" else k]) < TOL, (k, planes)
    if not split:
        return
    for scale in (1e4, 1e6, 1e-6):
        TOL = 1e-6          # fp32 level: tiles that took the second pass carry plain fp32-MFMA accumulation (the fp32 tiles' own bar), the others
                            # at most the floor terms the L_a + L_b bar admits
        # (a) one token of the activations and of the incoming gradient (rows of A in NT / NN; a term of every sum in TN)
        xa, dya = x.clone(), dy.clone()
        xa[5] *= scale; dya[7] *= scale
        for planes in (False, True):
            out, _, redone = products(xa, w, dya, planes)
            r = refs(xa, w, dya)
            # (1e4 = 2^13.3 sits at the edge of the tensor-wide scale's range: either pass will do; ONE row 10^6 times smaller than the rest
            # is outvoted -- the tile is redone when an eighth of its rows see the spread -- and keeps the absolute floor, 2^-39 of the
            # tensor's largest magnitude: only that row's own, tiny, output sees it)
            assert redone > 0 or scale <= 1e4
            assert rel(others(out["nt"].cpu(), 0, 5), others(r["nt"], 0, 5)) < TOL and rel(others(out["nn"].cpu(), 0, 7), others(r["nn"], 0, 7)) < TOL, (scale, planes)
            if scale > 1:
                assert rel(out["nt"][5], r["nt"][5]) < TOL and rel(out["nn"][7], r["nn"][7]) < TOL, (scale, planes)
            for k in ("tn", "tng"):
                if k in out:
                    assert rel(out[k], r["tn"]) < TOL, (k, scale, planes)
        # (b) one CHANNEL of the activations / of the gradient (a term of every sum in NT / NN; a row / column of the weight gradient)
        xb, dyb = x.clone(), dy.clone()
        xb[:, 9] *= scale; dyb[:, 11] *= scale
        out, _, redone = products(xb, w, dyb, True)
        r = refs(xb, w, dyb)
        assert rel(out["nt"], r["nt"]) < TOL and rel(others(out["nn"].cpu(), 0, []), r["nn"]) < TOL, scale
        for k in ("tn", "tng"):
            if k in out:
                o, rr = out[k].cpu(), r["tn"]
                assert rel(others(others(o, 0, 11), 1, 9), others(others(rr, 0, 11), 1, 9)) < TOL, (k, scale)      # every ordinary entry
                # the scaled row and column without their crossing, and the crossing itself -- ONE sum of 8 192 terms of both signs, so
                # measured against the size of its terms (a relative bar on a single cancelling sum is a lottery in any arithmetic)
                # (a single channel 10^6 times SMALLER than its three neighbours in a row-contiguous operand's group of four is the one
                # pattern the tracking does not see: that row of the weight gradient keeps the absolute floor, 2^-39 of the tensor's largest)
                tol_small = TOL if scale > 1 else 2e-5
                assert rel(others(o[11], 0, 9), others(rr[11], 0, 9)) < tol_small and rel(others(o[:, 9], 0, 11), others(rr[:, 9], 0, 11)) < tol_small, (k, scale)
                cross = float(dyb[:Mk, 11].double().norm() * xb[:Mk, 9].double().norm())
                assert abs(float(o[11, 9]) - float(rr[11, 9])) < (3e-8 if scale > 1 else 1e-6) * cross, (k, scale)
        # (c) one output feature (row) and one input feature (column) of the WEIGHT
        wc = w.clone()
        wc[3] *= scale; wc[:, 13] *= scale
        for planes in (False, True):
            out, _, redone = products(x, wc, dy, planes)
            r = refs(x, wc, dy)
            assert redone > 0 or scale <= 1e4
            assert rel(others(out["nt"].cpu(), 1, 3), others(r["nt"], 1, 3)) < TOL and rel(others(out["nn"].cpu(), 1, 13), others(r["nn"], 1, 13)) < TOL, (scale, planes)
            if scale > 1:          # (one dead output / input feature of a weight is outvoted, as a single tiny row of the activations is)
                assert rel(out["nt"][:, 3], r["nt"][:, 3]) < TOL and rel(out["nn"][:, 13], r["nn"][:, 13]) < TOL, (scale, planes)
    # a stale word (10^4 x too small: the pieces would overflow fp16) is caught by the same tracking
    xd, wd = x.cuda(), w.cuda()
    ops.begin_pass()
    word = ops.amax_for(xd)
    big = (x * 1e4).cuda()
    big._gaot_amax = (word, big._version, ops._PASS_ID[0])
    y = ops.linear_nt(big, wd)
    assert torch.isfinite(y).all() and rel(y, (x.double() * 1e4) @ w.double().t()) < TOL


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(8192, 256, 256), (4096, 2048, 256), (4096, 768, 256), (1000, 200, 96), (8192, 256, 1024), (300, 136, 32)])
def test_all_dma_tiles_equal_the_staged_tiles_bit_for_bit(M, N, K):
    """gemm_ad.hip (both operands global -> LDS by DMA, A split in registers from its fp32 LDS image) forms the SAME piece products in the
    SAME order as the staged tiles of gemm_split.hip: x W^T, dY W, the epilogues (bias + residual, SwiGLU gate with its saved pre-activations,
    a concatenated input), on 64- and 128-row tiles -- equal to the last bit, through the library's A/B switch."""
    from gaot_amd import ops, _lib
    lib = _lib.load()
    assert ops.precision() == "f32" and ops._F16_PIECES[0]
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).cuda()
    dy = torch.randn(M, N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).cuda()
    w = torch.nn.Parameter((torch.randn(N, K, generator=g) * 0.06).cuda())
    wcat = torch.nn.Parameter((torch.randn(N, 2 * K, generator=g) * 0.06).cuda())
    b = torch.randn(N, generator=g).cuda()
    w1 = torch.nn.Parameter((torch.randn(N, K, generator=g) * 0.06).cuda())
    w3 = torch.nn.Parameter((torch.randn(N, K, generator=g) * 0.06).cuda())
    w2 = torch.nn.Parameter((torch.randn(K, N, generator=g) * 0.06).cuda())

    def run(mode):
        old = lib.gaot_debug_set_gemm_ad(mode)
        try:
            ops.begin_pass()
            ops.adopt_adjacent_storage([w1, w3])
            ops.refresh_weight_amax([w, wcat, w1, w3, w2])
            out = {}
            with torch.no_grad():
                out["nt"] = ops.linear_nt(x, w.detach())
                out["path"] = lib.gaot_debug_last_gemm_path()
                out["nn"] = ops.matmul_nn(dy, w.detach())
                out["lin"] = ops.linear(x, w, b, residual=res)
                if K % 32 == 0:
                    out["cat"] = ops.linear(x, wcat, b, x2=x * 0.5)
                if N % 64 == 0 and K % 32 == 0:
                    out["ffn"] = ops.swiglu_ffn(x, w1, w3, w2)
            torch.cuda.synchronize()
            return out
        finally:
            lib.gaot_debug_set_gemm_ad(old)

    ref = run(0)
    if ref.pop("path") != 3:
        pytest.skip("this product does not run on the fp16-piece tiles")
    for mode in (1, 2, 3):
        got = run(mode)
        got.pop("path")
        for k, v in got.items():
            assert torch.equal(v, ref[k]), (mode, k, float((v - ref[k]).abs().max()))
    if M >= 8000:          # the 64 x 64 tiles (off by default; at fewer rows their switch also moves products off the fp32-MFMA tiles)
        old = lib.gaot_debug_set_gemm_ad_narrow(3)
        try:
            ops._PATH_CACHE.clear()
            got = run(1)
        finally:
            lib.gaot_debug_set_gemm_ad_narrow(old)
            ops._PATH_CACHE.clear()
        got.pop("path")
        for k, v in got.items():
            assert torch.equal(v, ref[k]), ("64x64", k, float((v - ref[k]).abs().max()))


@pytest.mark.gpu
def test_captured_graph_does_not_bake_in_a_warmup_magnitude_word():
    """A persistent input buffer (TrainStep._x, autograph's e.p, the rollout's static x) is the SAME tensor object, at the same
    `_version`, in the eager warm-up pass and in the capture that follows.  A magnitude word remembered on it by the warm-up must not
    satisfy the captured pass: the graph would then hold no absmax launch for it and every replay would scale the fp16 pieces by the
    warm-up batch's magnitude -- a 100x larger batch overflows fp16 (inf / NaN), a smaller one loses precision.  Words are tagged with
    the pass that made them (ops._PASS_ID) and ignored by later passes."""
    from gaot_amd import ops, _lib
    lib = _lib.load()
    assert ops.precision() == "f32" and ops._F16_PIECES[0]
    g = torch.Generator().manual_seed(3)
    M, K, N = 8192, 64, 256
    x0, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1
    xs, wd = x0.cuda(), w.cuda()                 # xs: the static input
    ops.begin_pass()
    ops.linear(xs, wd)                           # eager warm-up: remembers a word on xs
    assert lib.gaot_debug_last_gemm_path() == 3 and getattr(xs, "_gaot_amax", None) is not None
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side, capture_error_mode="thread_local"):
        ops.begin_pass()
        y = ops.linear(xs, wd)
    for scale in (100.0, 1e-3, 1e6):
        xs.copy_(x0.cuda() * scale)
        gr.replay()
        torch.cuda.synchronize()
        ref = (x0.double() * scale) @ w.double().t()
        assert torch.isfinite(y).all(), scale
        assert rel(y, ref) < 6e-7, (scale, rel(y, ref))
    # inference tensors have no version counter: publishing / looking up their words must not raise
    with torch.inference_mode():
        ops.begin_pass()
        xi = (x0.cuda() * 3.0)
        yi = ops.linear(ops.linear(xi, wd), torch.randn(64, N, generator=g).cuda())
        assert torch.isfinite(yi).all()


@pytest.mark.gpu
def test_gemm_split_bf16_exactness_of_the_split():
    """x = x1 + x2 + x3 is exact, so products of values with <= 8 significant bits are reproduced EXACTLY, and
    integer-valued operands (|sum| < 2^24) give the exact integer result."""
    from gaot_amd import ops, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    M, N, K = 256, 256, 64
    a = torch.randint(-2047, 2048, (M, K), generator=g).float()        # 12 significant bits: needs two pieces
    b = torch.randint(-15, 16, (N, K), generator=g).float()
    old = lib.gaot_debug_set_gemm_glds(5)
    try:
        out = ops.linear_nt(a.cuda(), b.cuda())
        assert lib.gaot_debug_last_gemm_path() == 3
    finally:
        lib.gaot_debug_set_gemm_glds(old)
    assert torch.equal(out.cpu().double(), a.double() @ b.double().t())


@pytest.mark.gpu
@pytest.mark.parametrize("E,cin,n,act,cout", [(55592, 4, 4, "gelu", 64), (1000, 4, 4, "gelu", 64), (131, 6, 4, "gelu", 64), (4097, 4, 3, "gelu", 64),
                                              (300, 12, 2, "gelu", 64), (31, 4, 4, "gelu", 64), (4096, 7, 3, "relu", 64), (16384, 7, 3, "relu", 64),
                                              (77, 9, 3, "relu", 64), (5000, 6, 4, "gelu", 48), (333, 4, 4, "gelu", 32), (1025, 6, 3, "gelu", 4),
                                              (700, 4, 2, "relu", 60), (4000, 9, 3, "relu", (64, 48, 48)), (999, 6, 4, "gelu", (32, 64, 16, 8)),
                                              (2049, 4, 3, "gelu", (48, 48, 48))])
def test_fused_kernel_mlp_matches_chain_and_float64(E, cin, n, act, cout):
    """csrc/kernel_mlp.hip (one launch forward, one backward) against the GEMM-chain path of the same op and a float64
    torch reference: output, every weight / bias gradient (mlp.py:307-337 semantics, exact-erf GELU).  cout < 64: a last layer
    narrower than the hidden width (lifting_channels = 48 at the 3-D configuration); a tuple: every width (zero-padded to 64 inside)."""
    from gaot_amd import ops
    torch.manual_seed(E + cin + n)
    d = "cuda"
    x = torch.rand(E, cin, device=d) * 2 - 1
    dims = [cin] + (list(cout) if isinstance(cout, tuple) else [64] * (n - 1) + [cout])      # a tuple: every layer's width
    cout = dims[-1]
    ws = [(torch.randn(dims[i + 1], dims[i], device=d) / dims[i] ** 0.5).requires_grad_() for i in range(n)]
    bs = [(0.1 * torch.randn(dims[i + 1], device=d)).requires_grad_() for i in range(n)]
    acts = [act] * (n - 1) + ["none"]
    dk = torch.randn(E, cout, device=d)
    assert ops._KernelMLP.eligible(x, ws, bs, acts)
    y = ops.mlp_chain(x, ws, bs, acts)
    gf = torch.autograd.grad(y, ws + bs, dk)
    old = ops._FUSED_KERNEL_MLP
    ops._FUSED_KERNEL_MLP = False
    try:
        y0 = ops.mlp_chain(x, ws, bs, acts)
        g0 = torch.autograd.grad(y0, ws + bs, dk)
    finally:
        ops._FUSED_KERNEL_MLP = old
    h = x.double()
    wd = [w.detach().double().requires_grad_() for w in ws]
    bd = [b.detach().double().requires_grad_() for b in bs]
    for i in range(n):
        h = h @ wd[i].t() + bd[i]
        if i < n - 1:
            h = torch.nn.functional.gelu(h) if act == "gelu" else torch.relu(h)
    gd = torch.autograd.grad(h, wd + bd, dk.double())
    assert rel(y, h) < 2e-6 and rel(y, y0) < 2e-6
    for a, b, c in zip(gf, g0, gd):
        assert a.shape == c.shape
        assert rel(a, c) < 1e-5, (rel(a, c), rel(b, c))
        assert rel(a, c) < 4 * rel(b, c) + 2e-6          # as accurate as the chain path



@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(8192, 2048, 256), (8192, 256, 1024), (4096, 768, 256)])
def test_gemm_two_piece_products(M, N, K, bf16x2):
    """gaot_gemm_desc.pieces = 2 (the opt-in "bf16x2" precision): every operand as two bf16 pieces, both rounded to nearest
    (16 significant bits, unbiased), three piece products.  On random normal data -- sums with full cancellation, the worst case for a
    per-term relative error -- the products stay within 6e-6 of float64 (measured 3-4.5e-6; the exact three-piece products: 2-5e-7), for
    all three product kinds and the grouped weight-gradient launch; the default (exact products, checked by the fixture) stays under 1e-6."""
    from gaot_amd import ops, _lib
    assert ops.precision() == "bf16x2"
    g = torch.Generator().manual_seed(M + N + K)
    x, w, dy = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(M, N, generator=g)
    xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
    ref = {"nt": x.double() @ w.double().t(), "nn": dy.double() @ w.double(), "tn": dy.double().t() @ x.double()}
    run = {"nt": lambda: ops.linear_nt(xd, wd), "nn": lambda: ops.matmul_nn(dyd, wd), "tn": lambda: ops.matmul_tn(dyd, xd)}
    for kind in ("nt", "nn", "tn"):
        e2 = rel(run[kind](), ref[kind])
        on_split_tiles = _lib.load().gaot_debug_last_gemm_path() == 3
        old = ops.set_gemm_pieces(3)
        try:
            e3 = rel(run[kind](), ref[kind])
        finally:
            ops.set_gemm_pieces(**old)
        assert e3 < 1e-6 and e2 < 6e-6, (kind, e2, e3)
        if on_split_tiles:
            assert e2 > 2 * e3, (kind, e2, e3)          # the field reaches the kernel (products on the fp32 MFMA ignore it)
    # the grouped launch
    out2, out3 = torch.empty(N, K, device="cuda"), torch.empty(N, K, device="cuda")
    ops.wgrad_launch([(dyd, N, xd, K, out2, K, None, N, K, M)])
    old = ops.set_gemm_pieces(3)
    try:
        ops.wgrad_launch([(dyd, N, xd, K, out3, K, None, N, K, M)])
    finally:
        ops.set_gemm_pieces(**old)
    assert rel(out3, ref["tn"]) < 1e-6 and 2 * rel(out3, ref["tn"]) < rel(out2, ref["tn"]) < 6e-6
    again = torch.empty_like(out2)
    ops.wgrad_launch([(dyd, N, xd, K, again, K, None, N, K, M)])
    assert torch.equal(again, out2)                      # deterministic (fixed-order slab sums)
    with pytest.raises(ValueError):
        ops.set_gemm_pieces(4)


@pytest.mark.gpu
def test_attention_fp16_piece_products():
    """the default attention for head_dim 32..64 (`pieces` = 4): Q / K / V as two fp16 pieces scaled through ONE magnitude word, dO through its
    own, P by 2^13, dS per 32 x 32 tile by the tile's own maximum; three piece products per k-step on the f16 MFMA.
      * random normal data at the bench shape (8-wave kernels) and at a small one (4-wave forward): output and gradients as close to
        float64 as the three-piece bf16 kernels (within 1.5x run to run; < 6e-7 / 1.2e-6);
      * operands 1e-3 / gradients 1e-9 times smaller, or 3 / 1e+6 times larger: errors at the same level (the words follow the data);
      * many equal tokens (same-signed accumulation): no drift; deterministic."""
    from gaot_amd import ops, _lib
    assert ops.precision() == "f32" and ops._F16_PIECES[0]

    def run(qkv, go, H, D):
        B, S, _ = qkv.shape
        r = qkv.clone().double().requires_grad_(True)
        q, k, v = [r[..., i * H * D:(i + 1) * H * D].reshape(B, S, H, D).transpose(1, 2) for i in range(3)]
        ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), -1) @ v).transpose(1, 2).reshape(B, S, H * D)
        ref.backward(go.double())
        outs = []
        for mode in ("fp16x2", "fp16x2", "bf16x3"):
            old = ops.set_f32_pieces(mode)
            try:
                d = qkv.to(dev()).requires_grad_(True)
                out = ops.attention(d, H, H, D)
                out.backward(go.to(dev()))
            finally:
                ops.set_f32_pieces(old)
            outs.append((out.detach(), d.grad))
        assert all(bool(torch.isfinite(t).all()) for o in outs for t in o), [bool(torch.isfinite(t).all()) for o in outs for t in o]
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), (rel(outs[0][0], outs[1][0].double().cpu()), rel(outs[0][1], outs[1][1].double().cpu()))           # deterministic
        return (rel(outs[0][0], ref), rel(outs[0][1], r.grad)), (rel(outs[2][0], ref), rel(outs[2][1], r.grad))

    g = torch.Generator().manual_seed(11)
    # (4 x 1 024, 5 x 1 000, 4 x 1 056 tokens x 8 heads of 32: 128 .. 255 workgroups of 256 keys -- the backward shares a key block's query
    # tiles between two workgroups and adds their dK / dV by atomics: two contributions per element, deterministic; odd tile counts and a
    # ragged last tile included)
    for (B, S, H, D) in ((8, 1024, 8, 32), (2, 333, 4, 32), (4, 1024, 8, 32), (5, 1000, 8, 32), (4, 1056, 8, 32), (1, 1024, 8, 48), (2, 333, 4, 36), (4, 2048, 8, 64)):      # head_dim 33..64: the DH = 64 kernels
        qkv, go = torch.randn(B, S, 3 * H * D, generator=g), torch.randn(B, S, H * D, generator=g)
        base = None
        for sq, sg in ((1.0, 1.0), (1e-3, 1e-9), (3.0, 1e6)):
            (eo, eg), (eo3, eg3) = run(qkv * sq if sq == 1.0 else qkv * sq, go * sg, H, D)
            assert eo < 1.5 * eo3 + 2e-8 and eg < 1.5 * eg3 + 2e-8, (B, S, sq, sg, eo, eg, eo3, eg3)
            assert sq > 1.0 or (eo < 6e-7 and eg < 1.2e-6), (B, S, sq, sg, eo, eg)      # (3 x larger q, k: a 9 x sharper softmax amplifies every kernel's rounding)
    base = (torch.randn(1, 1, 3 * 8 * 32, generator=g) * 0.6).repeat(1, 2048, 1)
    idx = torch.randperm(2048, generator=g)[:200]
    base[0, idx] = torch.randn(200, 3 * 8 * 32, generator=g) * 0.6
    (eo, eg), _ = run(base + 1e-3 * torch.randn(1, 2048, 3 * 8 * 32, generator=g), torch.randn(1, 2048, 8 * 32, generator=g), 8, 32)
    assert eo < 2e-6 and eg < 1e-5, (eo, eg)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(4096, 384, 1152), (4096, 256, 2048), (4096, 384, 1056), (3200, 384, 1536)])
def test_gemm_narrow_output_long_reduction_in_one_pass(M, N, K):
    """narrow outputs (two or three 128-wide tiles across, half-filled launches) with a reduction of 1 025 .. 2 048 and a weight with planes:
    ONE launch of the 64 x 64 all-DMA tiles, which flush their accumulators every 1 024 values of k (gemm_ad.hip FL) -- the 3-D configuration's
    q|k|v input gradient, 4 096 x 384 x 1 152, ran as two K slabs on the fp32-MFMA tiles + a reduce.  Both layouts (x W^T, dY W) against float64
    and against the K-slab path; same-signed operands (the accumulator drift the cap exists for) stay at fp32 level"""
    from gaot_amd import ops, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    for positive in (False, True):
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) * 0.05
        dy = torch.randn(M, K, generator=g)            # dY of a Linear whose weight is w2 [K, N]
        w2 = torch.randn(K, N, generator=g) * 0.05
        if positive:
            x, w, dy, w2 = x.abs(), w.abs(), dy.abs(), w2.abs()
        res = {}
        for on in (1, 0):
            old = lib.gaot_debug_set_gemm_ad_flush(on)
            try:
                ops._PATH_CACHE.clear()
                ops.begin_pass()
                wd, w2d = torch.nn.Parameter(w.cuda()), torch.nn.Parameter(w2.cuda())
                ops.refresh_weight_amax([wd, w2d])
                with torch.no_grad():
                    nt = ops.linear_nt(x.cuda(), wd.detach())
                    p_nt = lib.gaot_debug_last_gemm_path()
                    nn = ops.matmul_nn(dy.cuda(), w2d.detach())
                    p_nn = lib.gaot_debug_last_gemm_path()
                torch.cuda.synchronize()
                res[on] = (nt, nn, p_nt, p_nn)
            finally:
                lib.gaot_debug_set_gemm_ad_flush(old)
                ops._PATH_CACHE.clear()
        r_nt, r_nn = x.double() @ w.double().t(), dy.double() @ w2.double()
        assert res[1][2] == 3 and res[1][3] == 3          # the fp16-piece family, in one launch
        for i, r in ((0, r_nt), (1, r_nn)):
            e1, e0 = rel(res[1][i], r), rel(res[0][i], r)
            assert e1 < 6e-7 and e1 < 1.5 * e0 + 5e-8, (positive, i, e1, e0)


@pytest.mark.gpu
@pytest.mark.parametrize("B,S,H,D", [(4, 1024, 8, 32), (5, 1000, 8, 32), (1, 4096, 8, 48), (3, 1500, 8, 36), (2, 2048, 8, 64)])
def test_attention_key_split_forward(B, S, H, D):
    """the key-split forward (gaot_attention_fwd_ws with a workspace: 128 .. 255 blocks of 256 queries x batch x heads -- the 4 x 1 024-token
    batch, the 3-D configuration's 1 x 4 096 x 8 x 48, ragged and odd tile counts): two workgroups per query block, each over half the key
    tiles, joined by attn_fwd_combine_kernel -- against float64 (output, and the gradients the backward forms from its log-sum-exp) and
    against the one-pass forward it replaces there: same error level, deterministic"""
    from gaot_amd import ops, _lib
    lib = _lib.load()
    on = 2 if D == 32 else 1          # (head_dim 32 is not taken by default -- its 4-wave kernel already runs two workgroups per CU -- but the kernel is the same template)
    assert lib.gaot_attention_fwd_workspace(B, S, H, D) == (0 if D == 32 else 2 * (B * S * H * D + B * H * S))
    g = torch.Generator().manual_seed(S + D + B)
    qkv, go = torch.randn(B, S, 3 * H * D, generator=g), torch.randn(B, S, H * D, generator=g)
    r = qkv.clone().double().requires_grad_(True)
    q, k, v = [r[..., i * H * D:(i + 1) * H * D].reshape(B, S, H, D).transpose(1, 2) for i in range(3)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), -1) @ v).transpose(1, 2).reshape(B, S, H * D)
    ref.backward(go.double())

    def run(ks):
        old = lib.gaot_debug_set_attention_keysplit(ks)
        try:
            assert (lib.gaot_attention_fwd_workspace(B, S, H, D) > 0) == bool(ks)
            d = qkv.to(dev()).requires_grad_(True)
            out = ops.attention(d, H, H, D)
            out.backward(go.to(dev()))
        finally:
            lib.gaot_debug_set_attention_keysplit(old)
        return out.detach(), d.grad

    (o2, g2), (o2b, g2b), (o1, g1) = run(on), run(on), run(0)
    assert torch.equal(o2, o2b) and torch.equal(g2, g2b)
    e2, e1 = rel(o2, ref), rel(o1, ref)
    assert e2 < 6e-7 and e2 < 1.5 * e1 + 2e-8, (e2, e1)
    eg2, eg1 = rel(g2, r.grad), rel(g1, r.grad)
    assert eg2 < 1.2e-6 and eg2 < 1.5 * eg1 + 2e-8, (eg2, eg1)


@pytest.mark.gpu
@pytest.mark.parametrize("B,S,H,Hkv,D", [(2, 2048, 8, 4, 64), (3, 1400, 8, 2, 48)])
def test_attention_key_split_and_query_split_with_grouped_kv_heads(B, S, H, Hkv, D):
    """both [r6] chip-filling paths of 32 < head_dim <= 64 (key-split forward, query-split 8-wave backward) with FEWER kv heads than query
    heads (attn.py:100-104 repeat_interleave): output and the gradient of the fused q | k | v projection against float64, and against the
    one-pass / 4-wave kernels"""
    from gaot_amd import ops, _lib
    lib = _lib.load()
    assert lib.gaot_attention_fwd_workspace(B, S, H, D) > 0
    g = torch.Generator().manual_seed(S + D)
    W = (H + 2 * Hkv) * D
    qkv, go = torch.randn(B, S, W, generator=g), torch.randn(B, S, H * D, generator=g)
    r = qkv.clone().double().requires_grad_(True)
    q = r[..., :H * D].reshape(B, S, H, D).transpose(1, 2)
    k = r[..., H * D:(H + Hkv) * D].reshape(B, S, Hkv, D).transpose(1, 2).repeat_interleave(H // Hkv, dim=1)
    v = r[..., (H + Hkv) * D:].reshape(B, S, Hkv, D).transpose(1, 2).repeat_interleave(H // Hkv, dim=1)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), -1) @ v).transpose(1, 2).reshape(B, S, H * D)
    ref.backward(go.double())

    def run(on):
        o1, o2 = lib.gaot_debug_set_attention_keysplit(on), lib.gaot_debug_set_attention_dh8(on)
        try:
            d = qkv.to(dev()).requires_grad_(True)
            out = ops.attention(d, H, Hkv, D)
            out.backward(go.to(dev()))
        finally:
            lib.gaot_debug_set_attention_keysplit(o1); lib.gaot_debug_set_attention_dh8(o2)
        return out.detach(), d.grad

    (oa, ga), (ob, gb) = run(1), run(0)
    assert rel(oa, ref) < 6e-7 and rel(oa, ref) < 1.5 * rel(ob, ref) + 2e-8
    assert rel(ga, r.grad) < 1.2e-6 and rel(ga, r.grad) < 1.5 * rel(gb, r.grad) + 2e-8, (rel(ga, r.grad), rel(gb, r.grad))


@pytest.mark.gpu
@pytest.mark.parametrize("B,S,H,D", [(2, 2048, 8, 64), (3, 1500, 8, 36), (1, 4096, 8, 48), (4, 2048, 8, 48)])
def test_attention_head_dim_64_eight_wave_backward(B, S, H, D):
    """32 < head_dim <= 64 on fp16 pieces: the 8-wave 256-key backward (attn_bwd_split8_dh_kernel<true, QS>) -- query-split between two
    workgroups at 128 .. 255 key blocks x batch x heads (the first three shapes: the 3-D configuration's 1 x 4 096 x 8 x 48 among them, a
    ragged last tile and an odd tile count in the second), whole key blocks per workgroup from 256 on (the last) -- against float64 and
    against the 4-wave 128-key kernel it replaces there: the same error level, deterministic, the published magnitude word bounds dK / dV"""
    from gaot_amd import ops, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(S + D)
    qkv, go = torch.randn(B, S, 3 * H * D, generator=g), torch.randn(B, S, H * D, generator=g)
    r = qkv.clone().double().requires_grad_(True)
    q, k, v = [r[..., i * H * D:(i + 1) * H * D].reshape(B, S, H, D).transpose(1, 2) for i in range(3)]
    ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), -1) @ v).transpose(1, 2).reshape(B, S, H * D)
    ref.backward(go.double())

    def run(dh8):
        old = lib.gaot_debug_set_attention_dh8(dh8)
        try:
            d = qkv.to(dev()).requires_grad_(True)
            out = ops.attention(d, H, H, D)
            out.backward(go.to(dev()))
        finally:
            lib.gaot_debug_set_attention_dh8(old)
        return d.grad

    g8, g8b, g4 = run(1), run(1), run(0)
    assert torch.equal(g8, g8b)                                   # deterministic (no atomics: the halves are added in a fixed order)
    e8, e4 = rel(g8, r.grad), rel(g4, r.grad)
    assert e8 < 1.2e-6 and e8 < 1.5 * e4 + 2e-8, (e8, e4)
    assert rel(g8, g4.double().cpu()) < 2e-6


@pytest.mark.gpu
def test_attention_backward_one_barrier_kernel_equals_the_two_barrier_one():
    """[r6] attn_bwd_h16_kernel<8> (one workgroup barrier per query tile, double-buffered tile planes and dQ partials, DPP reductions, the second
    pieces through v_fma_mix) computes what attn_bwd_split8_kernel<2, 2, true, true> does, in the same order: dQ, dK and dV are bit-identical --
    full tiles, a ragged last query tile / key block (S = 1 000, 1 056 + 8), GQA, tiny and huge operands -- and within 1.2e-6 of float64.  The
    4-wave form (two workgroups per CU, twice the dQ slabs) has bit-identical dK / dV and a dQ within fp32 rounding.  Shapes below the fused
    kernel's reach (fewer than 256 key blocks) are untouched by the switch."""
    from gaot_amd import ops, _lib
    lib = _lib.load()
    assert ops.precision() == "f32" and ops._F16_PIECES[0]
    g = torch.Generator().manual_seed(21)

    def grads(qkv, go, H, Hkv, mode):
        old = lib.gaot_debug_set_attention_h16(mode)
        try:
            ops.begin_pass()
            d = qkv.clone().requires_grad_(True)
            out = ops.attention(d, H, Hkv, 32)
            out.backward(go)
        finally:
            lib.gaot_debug_set_attention_h16(old)
        return d.grad

    assert lib.gaot_debug_set_attention_h16(0x18) == 0x18          # the default
    for (B, S, H, Hkv, sq, sg) in ((8, 1024, 8, 8, 1.0, 1.0), (8, 1000, 8, 8, 1.0, 1.0), (9, 1064, 8, 4, 1.0, 1.0), (8, 2048, 8, 8, 1e-3, 1e-9), (16, 512, 8, 8, 3.0, 1e6)):
        qkv = (torch.randn(B, S, (H + 2 * Hkv) * 32, generator=g) * sq).to(dev())
        go = (torch.randn(B, S, H * 32, generator=g) * sg).to(dev())
        want = grads(qkv, go, H, Hkv, 0)
        for mode in (0x18, 0x08):
            got = grads(qkv, go, H, Hkv, mode)
            assert torch.equal(got, want), (B, S, hex(mode), float((got - want).abs().max()))
        got4 = grads(qkv, go, H, Hkv, 4)
        nq = H * 32
        assert torch.equal(got4[..., nq:], want[..., nq:]) and rel(got4[..., :nq], want[..., :nq].double().cpu()) < 2e-7, (B, S)
        if sq == 1.0:
            r = qkv.double().cpu().requires_grad_(True)
            q = r[..., :nq].reshape(B, S, H, 32).transpose(1, 2)
            k, v = [r[..., nq + i * Hkv * 32:nq + (i + 1) * Hkv * 32].reshape(B, S, Hkv, 32).transpose(1, 2).repeat_interleave(H // Hkv, dim=1) for i in range(2)]
            ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(32), -1) @ v).transpose(1, 2).reshape(B, S, nq)
            ref.backward(go.double().cpu())
            assert rel(want, r.grad) < 1.2e-6, (B, S, rel(want, r.grad))


@pytest.mark.gpu
def test_attention_two_piece_variant(bf16x2):
    """the "bf16x2" attention (`pieces` = 2 per call): P / dS and the Q / K / V / dO operands as two rounded bf16 pieces (three piece products per
    k-step in all seven products).  Random normal data (the worst case for a per-term relative error): output within 1e-5 and
    gradients within 1.5e-5 of float64 (measured 6.6e-6 / 7.9e-6; exact splits: 2.5e-7 / 4.8e-7); many equal tokens (same-signed
    accumulation): no drift.  No debug override is active (-1: the kernels follow the call's argument)."""
    from gaot_amd import ops, _lib
    lib = _lib.load()
    assert lib.gaot_debug_set_attention_p_pieces(-1) == -1 and lib.gaot_debug_set_attention_operand_pieces(-1) == -1
    def run(qkv, go, H, D):
        B, S, _ = qkv.shape
        r = qkv.clone().double().requires_grad_(True)
        q, k, v = [r[..., i * H * D:(i + 1) * H * D].reshape(B, S, H, D).transpose(1, 2) for i in range(3)]
        ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), -1) @ v).transpose(1, 2).reshape(B, S, H * D)
        ref.backward(go.double())
        d = qkv.to(dev()).requires_grad_(True)
        out = ops.attention(d, H, H, D)
        out.backward(go.to(dev()))
        d2 = qkv.to(dev()).requires_grad_(True)
        out2 = ops.attention(d2, H, H, D)
        out2.backward(go.to(dev()))
        assert torch.equal(out, out2) and torch.equal(d.grad, d2.grad)           # deterministic
        return rel(out, ref), rel(d.grad, r.grad)
    g = torch.Generator().manual_seed(7)
    B, S, H, D = 8, 1024, 8, 32                           # the bench shape
    eo, eg = run(torch.randn(B, S, 3 * H * D, generator=g), torch.randn(B, S, H * D, generator=g), H, D)
    assert 1e-6 < eo < 1e-5 and 1e-6 < eg < 1.5e-5, (eo, eg)
    eo, eg = run(torch.randn(1, 512, 3 * 4 * 48, generator=g), torch.randn(1, 512, 4 * 48, generator=g), 4, 48)      # head_dim 48: the DH = 64 kernels
    assert 1e-6 < eo < 1e-5 and 1e-6 < eg < 1.5e-5, (eo, eg)
    old = lib.gaot_debug_set_attention_split(2)           # ... and their 8-wave / 256-key backward (picked when it fills the chip), ragged S
    try:
        eo, eg = run(torch.randn(2, 333, 3 * 4 * 36, generator=g), torch.randn(2, 333, 4 * 36, generator=g), 4, 36)
    finally:
        lib.gaot_debug_set_attention_split(old)
    assert 1e-6 < eo < 1e-5 and 1e-6 < eg < 1.5e-5, (eo, eg)
    base = (torch.randn(1, 1, 3 * H * D, generator=g) * 0.6).repeat(1, 2048, 1)
    idx = torch.randperm(2048, generator=g)[:200]
    base[0, idx] = torch.randn(200, 3 * H * D, generator=g) * 0.6
    eo, eg = run(base + 1e-3 * torch.randn(1, 2048, 3 * H * D, generator=g), torch.randn(1, 2048, H * D, generator=g), H, D)
    assert eo < 1e-5 and eg < 5e-5, (eo, eg)


@pytest.mark.gpu
@pytest.mark.parametrize("E,cin,n", [(55592, 4, 4), (5000, 6, 3), (400000, 4, 4), (600000, 4, 3)])      # 6e5 edges: more than 16 tiles per workgroup (accumulator flush)
def test_kernel_mlp_two_piece_variant(E, cin, n, bf16x2):
    """the "bf16x2" kernel MLP behind GELU: two rounded bf16 pieces per operand in the forward chain, the recompute, the
    input-gradient chain and (through bf16 planes in LDS) the weight gradient.  Random data: output within 1e-5, every parameter
    gradient within 1.5e-5 of float64 (measured 5.6e-6 / 7.6e-6 .. 9.8e-6 at 4e5 edges).  Behind ReLU the products stay EXACT
    (gates): same bars as the three-piece test."""
    from gaot_amd import ops
    assert ops._PIECES["kmlp"] == 2
    torch.manual_seed(E + n)
    d = "cuda"
    x = torch.rand(E, cin, device=d) * 2 - 1
    dims = [cin] + [64] * n
    for act, bar_out, bar_grad in (("gelu", 1e-5, 1.5e-5), ("relu", 2e-6, 2e-6)):
        ws = [(torch.randn(dims[i + 1], dims[i], device=d) / dims[i] ** 0.5).requires_grad_() for i in range(n)]
        bs = [(0.1 * torch.randn(64, device=d)).requires_grad_() for _ in range(n)]
        acts = [act] * (n - 1) + ["none"]
        dk = torch.randn(E, 64, device=d)
        y = ops.mlp_chain(x, ws, bs, acts)
        g = torch.autograd.grad(y, ws + bs, dk)
        h = x.double()
        wd = [w.detach().double().requires_grad_() for w in ws]
        bd = [b.detach().double().requires_grad_() for b in bs]
        for i in range(n):
            h = h @ wd[i].t() + bd[i]
            if i < n - 1:
                h = torch.nn.functional.gelu(h) if act == "gelu" else torch.relu(h)
        gd = torch.autograd.grad(h, wd + bd, dk.double())
        errs = [rel(a, b) for a, b in zip(g, gd)]
        if act == "gelu":
            assert rel(y, h) < bar_out and max(errs) < bar_grad, (act, rel(y, h), errs)
            assert rel(y, h) > 1e-6          # the two-piece kernels did run
            g_again = torch.autograd.grad(ops.mlp_chain(x, ws, bs, acts), ws + bs, dk)
            assert all(torch.equal(a, b) for a, b in zip(g, g_again))         # deterministic (per-workgroup partial rows, fixed-order sums)
        else:                                # exact products whatever the setting: bit-identical to the three-piece mode
            old = ops.set_precision("f32")
            try:
                y3 = ops.mlp_chain(x, ws, bs, acts)
                g3 = torch.autograd.grad(y3, ws + bs, dk)
            finally:
                ops.set_gemm_pieces(**old)
            assert rel(y, h) < bar_out and torch.equal(y, y3) and all(torch.equal(a, b) for a, b in zip(g, g3))

@pytest.mark.gpu
def test_branch_free_erf_accuracy():
    """common.h erf_nb (single-range 1 - 2^(-|x| Q(|x|)), no branch) behind every GELU of the path: absolute error of
    the GELU value against float64 stays at the fp32 rounding level over the whole range, including tiny and huge inputs."""
    from gaot_amd import ops, _lib
    xs = torch.cat([torch.linspace(-9, 9, 8192 * 4 - 4096), torch.logspace(-8, 0, 2048), -torch.logspace(-8, 0, 2048)])
    x = xs.reshape(8192, 4).contiguous().cuda()
    eye = torch.eye(4, device="cuda")
    z = torch.empty_like(x)
    y = ops.linear_nt(x, eye, act=_lib.ACT_GELU, aux_out=z, ld_aux=4)
    xd = x.double().cpu()
    ref = 0.5 * xd * (1.0 + torch.erf(xd * 0.70710678118654752440))
    err = (y.double().cpu() - ref).abs()
    assert float((err / (xd.abs() + 1e-30)).max()) < 1.2e-7          # |gelu err| <= 0.5 |x| * 2.1e-7
    # derivative through the backward epilogue: dx = g * gelu'(z)
    g = torch.ones_like(x)
    dx = ops.matmul_nn(g, eye, act=_lib.ACT_GELU_BWD, aux_in=z, ld_aux=4)
    cdf = 0.5 * (1.0 + torch.erf(xd * 0.70710678118654752440))
    dref = cdf + xd * torch.exp(-0.5 * xd * xd) * 0.3989422804014327
    assert float((dx.double().cpu() - dref).abs().max()) < 4e-7


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(8, 16384, 1), (3, 1001, 2), (1, 5, 1)])
def test_mse_loss_kernel(shape):
    """nn.MSELoss() (mean) as two small launches forward, one backward, with a device-scalar upstream gradient."""
    from gaot_amd import ops
    torch.manual_seed(sum(shape))
    p = torch.randn(*shape, device="cuda", requires_grad=True)
    y = torch.randn(*shape, device="cuda")
    loss = ops.mse_loss(p, y)
    (g,) = torch.autograd.grad(loss * 3.0, p)
    pd, yd = p.detach().double().cpu(), y.double().cpu()
    assert abs(float(loss) - float(((pd - yd) ** 2).mean())) < 1e-6 * float(((pd - yd) ** 2).mean())
    assert rel(g, 3.0 * 2.0 * (pd - yd) / pd.numel()) < 1e-6
    assert float(ops.mse_loss(p, y)) == float(loss)          # deterministic


@pytest.mark.gpu
@pytest.mark.parametrize("B,n_src,Q,ci,C", [(8, 2000, 500, 1, 64), (3, 300, 200, 4, 64), (1, 50, 40, 2, 32), (5, 400, 64, 3, 128)])
def test_lifted_gno_transform_matches_unfused(B, n_src, Q, ci, C):
    """Encoder transform with the linear lifting folded in (gno.hip lift_* kernels) against the same op on the materialised
    lifted features (magno.py:334 + agno.py:198,245-271): output, dk, dWl, dbl; empty segments included."""
    from gaot_amd import ops
    from gaot_amd.plan import GeometryPlan
    g = torch.Generator().manual_seed(B * 1000 + Q + ci)
    deg = torch.randint(0, 9, (Q,), generator=g)
    deg[::7] = 0                                                     # empty segments
    splits = torch.cat([torch.zeros(1, dtype=torch.long), deg.cumsum(0)])
    E = int(splits[-1])
    index = torch.randint(0, n_src, (E,), generator=g)
    d = "cuda"
    plan = GeometryPlan(index.to(d), splits.to(d), n_src)
    k = torch.randn(E, C, generator=g).to(d).requires_grad_()
    pn = torch.randn(B, n_src, ci, generator=g).to(d)
    wl = (torch.randn(C, ci, 1, generator=g) * 0.5).to(d).requires_grad_()
    bl = torch.randn(C, generator=g).to(d).requires_grad_()
    a = torch.rand(max(E, 1), generator=g).to(d)
    dout = torch.randn(B, Q, C, generator=g).to(d)
    assert ops._GNOLiftTransform.eligible(pn, wl, C, a)
    y = ops.gno_lift_transform(k, pn, wl, bl, plan, a)
    gf = torch.autograd.grad(y, [k, wl, bl], dout)
    y0 = ops.gno_transform(k, ops.linear(pn, wl, bl), plan, a)
    g0 = torch.autograd.grad(y0, [k, wl, bl], dout)
    kd, wd, bd = k.detach().double().cpu().requires_grad_(), wl.detach().double().cpu().requires_grad_(), bl.detach().double().cpu().requires_grad_()
    f = pn.double().cpu() @ wd.reshape(C, ci).t() + bd
    eq = torch.repeat_interleave(torch.arange(Q), deg)
    contrib = a[:E].double().cpu()[None, :, None] * kd[None] * f[:, index, :]
    yd = torch.zeros(B, Q, C, dtype=torch.float64).index_add_(1, eq, contrib)
    gd = torch.autograd.grad(yd, [kd, wd, bd], dout.double().cpu())
    assert rel(y, yd) < 2e-6 and rel(y0, yd) < 2e-6
    for u, v, w in zip(gf, g0, gd):
        assert u.shape == w.shape
        assert rel(u, w) < 1e-5, (rel(u, w), rel(v, w))


@pytest.mark.gpu
@pytest.mark.parametrize("B,n_src,Q,OC,C,rb,bias", [(8, 500, 2000, 1, 64, True, False), (3, 200, 300, 4, 64, False, True),
                                                    (1, 40, 50, 2, 32, True, True), (5, 64, 400, 3, 128, False, False)])
def test_projected_gno_transform_matches_unfused(B, n_src, Q, OC, C, rb, bias):
    """Decoder transform with the trailing point-wise linear maps folded in (gno.hip proj_* kernels) against transform + GEMM
    (magno.py:640-668 after agno.py:245-271): y, dk, dF, dweff, d rowbias, d bias; empty segments included."""
    from gaot_amd import ops
    from gaot_amd.plan import GeometryPlan
    g = torch.Generator().manual_seed(B * 1000 + Q + OC)
    deg = torch.randint(0, 7, (Q,), generator=g)
    deg[::5] = 0
    splits = torch.cat([torch.zeros(1, dtype=torch.long), deg.cumsum(0)])
    E = int(splits[-1])
    index = torch.randint(0, n_src, (E,), generator=g)
    d = "cuda"
    plan = GeometryPlan(index.to(d), splits.to(d), n_src)
    k = torch.randn(E, C, generator=g).to(d).requires_grad_()
    f = torch.randn(B, n_src, C, generator=g).to(d).requires_grad_()
    weff = (torch.randn(OC, C, generator=g) * 0.3).to(d).requires_grad_()
    rowb = torch.randn(Q, OC, generator=g).to(d).requires_grad_() if rb else None
    bs = torch.randn(OC, generator=g).to(d).requires_grad_() if bias else None
    a = torch.rand(max(E, 1), generator=g).to(d)
    dy = torch.randn(B, Q, OC, generator=g).to(d)
    params = [k, f, weff] + ([rowb] if rb else []) + ([bs] if bias else [])
    assert ops._GNOProjTransform.eligible(f, weff, a)
    y = ops.gno_proj_transform(k, f, weff, rowb, bs, plan, a)
    gf = torch.autograd.grad(y, params, dy)
    y0 = ops.linear(ops.gno_transform(k, f, plan, a), weff, bs, rowbias=rowb)
    g0 = torch.autograd.grad(y0, params, dy)
    pd = [p.detach().double().cpu().requires_grad_() for p in params]
    kd, fd, wd = pd[:3]
    eq = torch.repeat_interleave(torch.arange(Q), deg)
    A = torch.zeros(B, Q, C, dtype=torch.float64).index_add_(1, eq, a[:E].double().cpu()[None, :, None] * kd[None] * fd[:, index, :])
    yd = A @ wd.t()
    i = 3
    if rb:
        yd = yd + pd[i][None]; i += 1
    if bias:
        yd = yd + pd[i]
    gd = torch.autograd.grad(yd, pd, dy.double().cpu())
    assert rel(y, yd) < 2e-6 and rel(y0, yd) < 2e-6
    for u, v, w in zip(gf, g0, gd):
        assert u.shape == w.shape
        assert rel(u, w) < 1e-5, (rel(u, w), rel(v, w))
    # the edge-partitioned dF kernels (batches of four and more) publish dF's magnitude word (gaot_gno_proj_gather_t_ep_w): exactly max |dF|
    aw = getattr(gf[1], "_gaot_amax", None)
    if B >= 4 and ops.wants_amax():
        assert aw is not None and float(aw[0].max()) == float(gf[1].abs().max())


@pytest.mark.parametrize("B,OC", [(8, 1), (5, 2)])
def test_projected_gno_transform_walks_rows_in_neighbour_order(B, OC):
    """a dict-cached plan (fx geometries) carries `row_order` (rows sorted by first source row): the batch-inside decoder forward walks
    its rows in that order, the edge-gradient kernel its edges in the transposed CSR's order, both in contiguous ranges per XCD -- the
    gathered feature rows stay in cache.  Per row / per edge the arithmetic is unchanged: y and dk are bit-identical to a walk in the
    caller's order, dF too; dweff (workgroup partials cut differently) to fp32 rounding; everything against float64."""
    from gaot_amd import ops
    from gaot_amd.plan import plan_for
    Q, n_src, C = 6000, 1024, 64
    g = torch.Generator().manual_seed(B + OC)
    deg = torch.randint(0, 7, (Q,), generator=g)
    deg[::7] = 0
    splits = torch.cat([torch.zeros(1, dtype=torch.long), deg.cumsum(0)])
    E = int(splits[-1])
    index = torch.randint(0, n_src, (E,), generator=g)
    d = "cuda"
    nb = {"neighbors_index": index.to(d), "neighbors_row_splits": splits.to(d)}
    plan = plan_for(nb, n_src)
    order = plan.row_order
    assert order is not None and sorted(order.tolist()) == list(range(Q))
    first = index[splits[:-1].clamp(max=E - 1)][order.cpu().long()]
    nonempty = int((deg > 0).sum())
    assert bool((first[:nonempty][1:] >= first[:nonempty][:-1]).all()) and bool((deg[order.cpu().long()][nonempty:] == 0).all())
    k = torch.randn(E, C, generator=g).to(d).requires_grad_()
    f = torch.randn(B, n_src, C, generator=g).to(d).requires_grad_()
    weff = (torch.randn(OC, C, generator=g) * 0.3).to(d).requires_grad_()
    rowb = torch.randn(Q, OC, generator=g).to(d).requires_grad_()
    a = torch.rand(E, generator=g).to(d)
    dy = torch.randn(B, Q, OC, generator=g).to(d)
    params = [k, f, weff, rowb]
    res = []
    for ident in (False, True):
        plan._row_order = torch.arange(Q, device=d, dtype=torch.int32) if ident else order
        y = ops.gno_proj_transform(k, f, weff, rowb, None, plan, a)
        res.append((y.detach(), torch.autograd.grad(y, params, dy)))
    plan._row_order = order
    assert torch.equal(res[0][0], res[1][0])
    for u, v, name in zip(res[0][1], res[1][1], ("dk", "df", "dweff", "drowb")):
        assert torch.equal(u, v), name
    pd = [p.detach().double().cpu().requires_grad_() for p in params]
    eq = torch.repeat_interleave(torch.arange(Q), deg)
    A = torch.zeros(B, Q, C, dtype=torch.float64).index_add_(1, eq, a.double().cpu()[None, :, None] * pd[0][None] * pd[1][:, index, :])
    yd = A @ pd[2].t() + pd[3][None]
    gd = torch.autograd.grad(yd, pd, dy.double().cpu())
    assert rel(res[0][0], yd) < 2e-6
    for u, w in zip(res[0][1], gd):
        assert rel(u, w) < 1e-5


def test_segment_csr_wrapper_matches_reference_semantics():
    """utils/segment_csr.py (the reference wrapper's signature, utils/segment_csr.py:14-55): sum / mean over CSR segments of
    [E], [E,C] and [B,E,C] inputs, empty segment -> 0, gradient = broadcast of the row gradient"""
    from gaot_amd.model.layers.utils.segment_csr import segment_csr
    g = torch.Generator().manual_seed(4)
    deg = torch.tensor([3, 0, 5, 1, 0, 7])
    ip = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(deg, 0)])
    qid = torch.repeat_interleave(torch.arange(6), deg)
    for shape in ((16,), (16, 8), (3, 16, 8)):
        x = torch.randn(*shape, generator=g)
        for red in ("sum", "mean"):
            xd = x.to(dev()).requires_grad_(True)
            out = segment_csr(xd, ip.to(dev()), reduce=red)
            dim = 0 if x.dim() < 3 else 1
            ref = torch.zeros(*([6] if x.dim() == 1 else ([6, 8] if x.dim() == 2 else [3, 6, 8])), dtype=torch.float64)
            ref.index_add_(dim, qid, x.double())
            if red == "mean":
                d = deg.clamp(min=1).double()
                ref = ref / (d if x.dim() == 1 else (d[:, None] if x.dim() == 2 else d[None, :, None]))
            assert rel(out, ref) < 1e-6 and out.shape == ref.shape
            w = torch.randn(out.shape, generator=g)
            (out * w.to(dev())).sum().backward()
            gref = w.double().index_select(dim, qid)
            if red == "mean":
                d = (1.0 / deg.clamp(min=1).double())[qid]
                gref = gref * (d if x.dim() == 1 else (d[:, None] if x.dim() == 2 else d[None, :, None]))
            assert rel(xd.grad, gref) < 1e-6
        # reduce='max' (torch_scatter's, used by the reference's segment softmax agno.py:131-133): per-segment maximum, empty -> 0
        xd = x.to(dev())
        out = segment_csr(xd, ip.to(dev()), reduce="max")
        xs = x.double() if x.dim() == 3 else (x.double()[None] if x.dim() == 2 else x.double()[None, :, None])
        ref = torch.zeros(xs.shape[0], 6, xs.shape[2], dtype=torch.float64)
        for q in range(6):
            if deg[q] > 0:
                ref[:, q] = xs[:, int(ip[q]):int(ip[q + 1])].max(dim=1).values
        ref = ref if x.dim() == 3 else (ref[0] if x.dim() == 2 else ref[0, :, 0])
        assert out.shape == ref.shape and torch.equal(out.double().cpu(), ref)
    with pytest.raises(ValueError):       # the reference's native branch (use_scatter=False) knows mean and sum only
        segment_csr(torch.zeros(4, device=dev()), torch.tensor([0, 4], device=dev()), reduce="max", use_scatter=False)
    with pytest.raises(ValueError):
        segment_csr(torch.zeros(4, device=dev()), torch.tensor([0, 4], device=dev()), reduce="prod")


def test_glue_kernels_against_float64():
    """dot-product edge score, segment max pool (even split among ties), multiscale mixing, conditioned-norm modulation, RoPE"""
    from gaot_amd import ops
    from gaot_amd.plan import GeometryPlan
    from oracle import gaot_oracle as O
    g = torch.Generator().manual_seed(6)
    x, lat, enc, _ = _random_geometry(6, 700, [12, 12], 0.3)
    plan = GeometryPlan(enc[0].to(dev()), enc[1].to(dev()), 700)
    idx, sp = enc
    qid, deg = O.edge_query_ids(sp)
    E, Q = idx.numel(), 144
    # dot score + its two node gradients
    qn, kn = torch.randn(Q, 64, generator=g), torch.randn(700, 64, generator=g)
    qd, kd = qn.to(dev()).requires_grad_(True), kn.to(dev()).requires_grad_(True)
    sc = ops.edge_dot_score(qd, kd, plan, 0.125)
    q64, k64 = qn.double().requires_grad_(True), kn.double().requires_grad_(True)
    ref = (q64[qid] * k64[idx]).sum(-1) * 0.125
    assert rel(sc[:E], ref) < 1e-6
    w = torch.randn(E, generator=g)
    (sc[:E] * w.to(dev())).sum().backward()
    (ref * w.double()).sum().backward()
    assert rel(qd.grad, q64.grad) < 1e-6 and rel(kd.grad, k64.grad) < 1e-6
    # segment max with ties (ReLU zeros) -> even split, as torch.scatter_reduce(amax)
    h = torch.relu(torch.randn(E, 16, generator=g))
    hd = h.to(dev()).requires_grad_(True)
    out = ops.segment_max(hd, plan)
    h64 = h.double().requires_grad_(True)
    ref = torch.zeros(Q, 16, dtype=torch.float64).scatter_reduce(0, qid[:, None].expand(E, 16), h64, reduce="amax", include_self=False)
    assert torch.equal(out.cpu().double(), ref.detach())
    w = torch.randn(Q, 16, generator=g)
    (out * w.to(dev())).sum().backward()
    (ref * w.double()).sum().backward()
    assert rel(hd.grad, h64.grad) < 1e-6
    # multiscale mixing
    ts = [torch.randn(3, Q, 24, generator=g) for _ in range(3)]
    wq = torch.softmax(torch.randn(Q, 3, generator=g), -1)
    tds = [t.to(dev()).requires_grad_(True) for t in ts]
    wd = wq.to(dev()).requires_grad_(True)
    t64 = [t.double().requires_grad_(True) for t in ts]
    w64 = wq.double().requires_grad_(True)
    o1, o2 = ops.scale_mix(tds, wd), ops.scale_mix(tds, None)
    r1 = sum(w64[None, :, i:i + 1] * t64[i] for i in range(3))
    r2 = torch.stack(t64, 0).mean(0)
    assert rel(o1, r1) < 1e-6 and rel(o2, r2) < 1e-6
    gw = torch.randn(3, Q, 24, generator=g)
    ((o1 + 0.5 * o2) * gw.to(dev())).sum().backward()
    ((r1 + 0.5 * r2) * gw.double()).sum().backward()
    assert rel(wd.grad, w64.grad) < 1e-5 and all(rel(a.grad, b.grad) < 1e-6 for a, b in zip(tds, t64))
    # conditioned-norm modulation
    xx, s_, b_ = torch.randn(3, 300, 48, generator=g), torch.randn(3, 48, generator=g), torch.randn(3, 48, generator=g)
    ds = [t.to(dev()).requires_grad_(True) for t in (xx, s_, b_)]
    d64 = [t.double().requires_grad_(True) for t in (xx, s_, b_)]
    y = ops.cond_affine(*ds)
    yr = d64[0] * d64[1][:, None, :] + d64[2][:, None, :]
    assert rel(y, yr) < 1e-6
    gy = torch.randn(3, 300, 48, generator=g)
    (y * gy.to(dev())).sum().backward()
    (yr * gy.double()).sum().backward()
    assert all(rel(a.grad, b.grad) < 2e-6 for a, b in zip(ds, d64))
    # RoPE: in place on the q|k part of a fused projection output; backward = transposed rotation
    B, S, H, Hkv, D = 2, 37, 4, 2, 16
    qkv = torch.randn(B, S, (H + 2 * Hkv) * D, generator=g)
    freqs = 1.0 / (10000.0 ** (torch.arange(0, D, 2).float() / D))
    ang = torch.arange(S, dtype=torch.float32)[:, None] * freqs[None, :]
    cs = torch.stack([ang.cos(), ang.sin()], -1).contiguous().to(dev())
    base = qkv.to(dev()).requires_grad_(True)
    out = ops.rope(base, H + Hkv, D, cs)
    q64 = qkv.double().requires_grad_(True)
    heads = q64.reshape(B, S, H + 2 * Hkv, D)
    rot = O.rotate_queries_or_keys(heads[:, :, :H + Hkv].transpose(1, 2), freqs.double()).transpose(1, 2)
    ref = torch.cat([rot, heads[:, :, H + Hkv:]], dim=2).reshape(B, S, -1)
    assert rel(out, ref) < 1e-6
    gq = torch.randn(B, S, (H + 2 * Hkv) * D, generator=g)
    (out * gq.to(dev())).sum().backward()
    (ref * gq.double()).sum().backward()
    assert rel(base.grad, q64.grad) < 1e-6


@pytest.mark.parametrize("skew", [True, False])
@pytest.mark.parametrize("B,ci,OC", [(8, 1, 1), (3, 3, 2), (1, 2, 4)])
def test_edge_partitioned_transforms_match_row_parallel_and_float64(skew, B, ci, OC):
    """csrc/gno_ep.hip (edge chunks + segmented reductions + carry fix-up; batch-inside decoder forward) against the row-parallel
    kernels and float64, on a degree-skewed airfoil-like cloud (rows of 300+ edges, thousands of empty rows, rows spanning
    several 32-edge chunks) and on a uniform one; forward and every gradient."""
    from gaot_amd import ops
    from gaot_amd.plan import GeometryPlan
    from oracle import gaot_oracle as O
    from tests._workloads import naca_points, grid
    g = torch.Generator().manual_seed(17 + B)
    N, C = 6000, 64
    lat = grid([48, 48])
    x = naca_points(N, g, 0.12) if skew else (torch.rand(N, 2, generator=g) * 2 - 1)
    enc, dec = O.radius_csr(x, lat, 0.045), O.radius_csr(lat, x, 0.045)
    Q = lat.shape[0]
    results = {}
    for mode in (0, 1):
        old = ops.set_gno_ep(mode)
        try:
            pe = GeometryPlan(enc[0].to(dev()), enc[1].to(dev()), N)
            pd = GeometryPlan(dec[0].to(dev()), dec[1].to(dev()), Q)
            if skew:
                assert pe.rows_skewed and pd.t_rows_skewed and pe.max_deg > 200
            gg = torch.Generator().manual_seed(5)
            k_e = torch.randn(pe.E, C, generator=gg).to(dev()).requires_grad_(True)
            k_d = torch.randn(pd.E, C, generator=gg).to(dev()).requires_grad_(True)
            pn = torch.randn(B, N, ci, generator=gg).to(dev())
            wl = torch.randn(C, ci, generator=gg).to(dev()).requires_grad_(True)
            bl = torch.randn(C, generator=gg).to(dev()).requires_grad_(True)
            a_e = torch.rand(max(pe.E, 1), generator=gg).to(dev())
            a_d = torch.rand(max(pd.E, 1), generator=gg).to(dev())
            f = torch.randn(B, Q, C, generator=gg).to(dev()).requires_grad_(True)
            weff = torch.randn(OC, C, generator=gg).to(dev()).requires_grad_(True)
            rowb = torch.randn(N, OC, generator=gg).to(dev()).requires_grad_(True)
            out_e = ops.gno_lift_transform(k_e, pn, wl, bl, pe, a_e)
            out_d = ops.gno_proj_transform(k_d, f, weff, rowb, None, pd, a_d)
            w1, w2 = torch.randn(out_e.shape, generator=gg).to(dev()), torch.randn(out_d.shape, generator=gg).to(dev())
            ((out_e * w1).sum() + (out_d * w2).sum()).backward()
            results[mode] = [t.detach().cpu().double() for t in (out_e, out_d, k_e.grad, wl.grad, bl.grad, k_d.grad, f.grad, weff.grad, rowb.grad)]
            if mode == 1:       # float64 reference of the two forwards
                qe, _ = O.edge_query_ids(enc[1]); qd, _ = O.edge_query_ids(dec[1])
                fe = pn.cpu().double() @ wl.detach().cpu().double().t() + bl.detach().cpu().double()
                ref_e = torch.zeros(B, Q, C, dtype=torch.float64).index_add_(1, qe, a_e.cpu().double()[:pe.E, None] * k_e.detach().cpu().double() * fe[:, enc[0]])
                td = torch.zeros(B, N, C, dtype=torch.float64).index_add_(1, qd, a_d.cpu().double()[:pd.E, None] * k_d.detach().cpu().double() * f.detach().cpu().double()[:, dec[0]])
                ref_d = td @ weff.detach().cpu().double().t() + rowb.detach().cpu().double()
                assert rel(out_e, ref_e) < 2e-6 and rel(out_d, ref_d) < 2e-6
        finally:
            ops.set_gno_ep(old)
    for a, b in zip(results[0], results[1]):
        assert float((a - b).norm() / max(float(b.norm()), 1e-30)) < 2e-6


def test_gemm_bf16_piece_mode_is_a_plain_bf16_gemm():
    """gaot_debug_set_gemm_pieces(1) (bench --dtype bf16 only): operands rounded to nearest-even bf16, fp32 accumulation.  Against
    float64 products of the ROUNDED operands the error is fp32 accumulation error; against the exact operands it is bf16-level
    (and the default 3-piece mode, re-selected afterwards, is back at fp32 level)."""
    from gaot_amd import ops, _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(8)
    M, N, K = 1024, 384, 256
    x, w, gy = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(M, N, generator=g)
    r = lambda t: t.bfloat16().double()
    old_mode = ops.set_gemm_mode(5)
    old = lib.gaot_debug_set_gemm_pieces(1)
    try:
        y = ops.linear_nt(x.to(dev()), w.to(dev()))
        assert lib.gaot_debug_last_gemm_path() == 3
        dx = ops.matmul_nn(gy.to(dev()), w.to(dev()))
        dw = ops.matmul_tn(gy.to(dev()), x.to(dev()))
        assert rel(y, r(x) @ r(w).t()) < 2e-6 and rel(dx, r(gy) @ r(w)) < 2e-6 and rel(dw, r(gy).t() @ r(x)) < 2e-6
        assert 5e-4 < rel(y, x.double() @ w.double().t()) < 1e-2
    finally:
        lib.gaot_debug_set_gemm_pieces(old)
    y3 = ops.linear_nt(x.to(dev()), w.to(dev()))
    ops.set_gemm_mode(old_mode)
    assert rel(y3, x.double() @ w.double().t()) < 2e-6


def test_agno_with_relu_kernel_mlp_matches_float64():
    """channel_mlp_non_linearity other than the default GELU (reference agno.py:71, mlp.py:311): ReLU runs on the same fused
    kernels; anything else is refused by name"""
    import torch.nn.functional as F
    from gaot_amd.model.layers.agno import AGNO
    x, lat, enc, _ = _random_geometry(11, 600, [12, 12], 0.2)
    idx, sp = enc
    torch.manual_seed(3)
    layer = AGNO(channel_mlp_layers=[4, 64, 64, 64], channel_mlp_non_linearity=F.relu, transform_type="linear", use_attn=False).to(dev())
    f = torch.randn(2, 600, 64)
    fd = f.to(dev()).requires_grad_(True)
    out = layer(y=x.to(dev()), x=lat.to(dev()), f_y=fd, neighbors=_dict(enc))
    go = torch.randn(out.shape)
    out.backward(go.to(dev()))
    # float64: k_e = MLP_relu([y_j, x_i]); out[b,q] = mean over the row of k_e * f[b, j]
    w = [p.detach().cpu().double() for p in layer.channel_mlp.parameters()]
    ws = [q.clone().requires_grad_(True) for q in w]
    qid = torch.repeat_interleave(torch.arange(sp.numel() - 1), sp[1:] - sp[:-1])
    h = torch.cat([x[idx], lat[qid]], -1).double()
    for i in range(0, len(ws), 2):
        h = h @ ws[i].t() + ws[i + 1]
        if i + 2 < len(ws):
            h = torch.relu(h)
    fr = f.double().requires_grad_(True)
    contrib = h[None] * fr[:, idx]
    ref = torch.zeros(2, sp.numel() - 1, 64, dtype=torch.float64).index_add(1, qid, contrib)
    ref = ref / (sp[1:] - sp[:-1]).clamp(min=1)[None, :, None]
    ref.backward(go.double())
    assert rel(out, ref) < 3e-6 and rel(fd.grad, fr.grad) < 3e-6
    for prm, r in zip(layer.channel_mlp.parameters(), ws):
        assert rel(prm.grad, r.grad) < 2e-5
    # any OTHER activation callable (mlp.py:311) -- and dropout (mlp.py:318-322) -- is served as well: one HIP GEMM per layer with the
    # caller's callable in between
    torch.manual_seed(3)
    layer2 = AGNO(channel_mlp_layers=[4, 32, 64], channel_mlp_non_linearity=F.silu, use_attn=False, coord_dim=2).to(dev())
    out2 = layer2(y=x.to(dev()), x=lat.to(dev()), f_y=f.to(dev()), neighbors=_dict(enc))
    w2 = [p.detach().cpu().double() for p in layer2.channel_mlp.parameters()]
    h2 = torch.cat([x[idx], lat[qid]], -1).double()
    h2 = F.silu(h2 @ w2[0].t() + w2[1]) @ w2[2].t() + w2[3]
    ref2 = torch.zeros(2, sp.numel() - 1, 64, dtype=torch.float64).index_add(1, qid, h2[None] * f.double()[:, idx])
    ref2 = ref2 / (sp[1:] - sp[:-1]).clamp(min=1)[None, :, None]
    assert rel(out2, ref2) < 3e-6


def test_mlp_chain_with_a_final_activation():
    """an MLP chain whose LAST layer carries GELU / ReLU too (ops.mlp_chain acts = [..., 'gelu']): the derivative of the final
    activation has no following product to ride on and runs as its own kernel (gaot_act_bwd); against float64"""
    from gaot_amd import ops
    g = torch.Generator().manual_seed(3)
    for act in ("gelu", "relu"):
        x = torch.randn(700, 24, generator=g)
        ws = [torch.randn(40, 24, generator=g) / 5, torch.randn(12, 40, generator=g) / 6]
        bs = [torch.randn(40, generator=g) / 3, torch.randn(12, generator=g) / 3]
        go = torch.randn(700, 12, generator=g)
        f = (lambda t: torch.nn.functional.gelu(t)) if act == "gelu" else torch.relu
        xd = x.double().requires_grad_()
        wd = [w.double().requires_grad_() for w in ws]
        bd = [b.double().requires_grad_() for b in bs]
        ref = f(f(xd @ wd[0].t() + bd[0]) @ wd[1].t() + bd[1])
        gref = torch.autograd.grad(ref, [xd] + wd + bd, go.double())
        xg = x.to(dev()).requires_grad_()
        wg = [w.to(dev()).requires_grad_() for w in ws]
        bg = [b.to(dev()).requires_grad_() for b in bs]
        y = ops.mlp_chain(xg, wg, bg, [act, act])
        got = torch.autograd.grad(y, [xg] + wg + bg, go.to(dev()))
        assert rel(y, ref) < 2e-6
        for a, b in zip(got, gref):
            assert rel(a, b) < 5e-6, (act, rel(a, b))


def test_channel_mlp_arbitrary_activation_and_dropout():
    """ChannelMLP / LinearChannelMLP with a non-default `non_linearity` callable and dropout (mlp.py:253-298, 311-337): the
    Linear / Conv1d(k=1) layers stay HIP GEMMs; in eval mode dropout is the identity, in training mode it is torch's"""
    import torch.nn.functional as F
    from gaot_amd.model.layers.mlp import ChannelMLP, LinearChannelMLP
    torch.manual_seed(0)
    x = torch.randn(3, 500, 12)
    for mod in (LinearChannelMLP([12, 40, 24], non_linearity=F.tanh, dropout=0.25), ChannelMLP(12, 24, 40, n_layers=3, non_linearity=F.silu, dropout=0.1)):
        mod = mod.to(dev()).eval()
        ws = [p.detach().cpu().double().reshape(p.shape[0], -1) if p.dim() > 1 else p.detach().cpu().double() for p in mod.parameters()]
        act = mod.non_linearity
        h = x.double()
        for i in range(0, len(ws), 2):
            h = h @ ws[i].t() + ws[i + 1]
            if i + 2 < len(ws):
                h = act(h)
        y = mod(x.to(dev())) if isinstance(mod, LinearChannelMLP) else mod.forward_channels_last(x.to(dev()))
        assert rel(y, h) < 3e-6
        mod.train()
        yt = mod(x.to(dev())) if isinstance(mod, LinearChannelMLP) else mod.forward_channels_last(x.to(dev()))
        assert yt.shape == y.shape and float((yt == 0).float().mean()) > 0.05          # dropout zeroed its share of the outputs


def test_attention_many_equal_tokens_long_sequence():
    """4 096 keys of which 90 % are one repeated token (the C5 latent grid: empty rows): nearly uniform softmax over rows of V that
    are alike -- thousands of same-signed increments per output.  The bf16 MFMA does not round its accumulator to nearest, so
    the split kernels keep every tile's product off the running sums; checked at fp32-MFMA accuracy against float64."""
    from gaot_amd import ops, _lib
    g = torch.Generator().manual_seed(12)
    B, S, H, D = 1, 4096, 2, 48
    W = 3 * H * D
    base = torch.randn(1, 1, W, generator=g) * 0.6
    qkv = base.repeat(B, S, 1)
    idx = torch.randperm(S, generator=g)[: S // 10]
    qkv[0, idx] = torch.randn(idx.numel(), W, generator=g) * 0.6
    qkv = qkv + 1e-3 * torch.randn(B, S, W, generator=g)
    go = torch.randn(B, S, H * D, generator=g)
    r = qkv.clone().double().requires_grad_(True)
    q = r[..., :H * D].reshape(B, S, H, D).transpose(1, 2)
    k = r[..., H * D:2 * H * D].reshape(B, S, H, D).transpose(1, 2)
    v = r[..., 2 * H * D:].reshape(B, S, H, D).transpose(1, 2)
    ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D), -1) @ v).transpose(1, 2).reshape(B, S, H * D)
    ref.backward(go.double())
    errs = {}
    for mode in (1, 0):
        old = _lib.load().gaot_debug_set_attention_split(mode)
        try:
            d = qkv.to(dev()).requires_grad_(True)
            out = ops.attention(d, H, H, D)
            out.backward(go.to(dev()))
        finally:
            _lib.load().gaot_debug_set_attention_split(old)
        errs[mode] = (rel(out, ref), rel(d.grad, r.grad))
    assert errs[1][0] < 2e-6 and errs[1][1] < 2e-5, errs
    assert errs[1][0] < 2 * errs[0][0] + 1e-7 and errs[1][1] < 2 * errs[0][1] + 1e-7, errs      # no worse than the fp32 MFMA


def test_split_gemm_same_signed_long_reduction():
    """all-positive operands, K = 4 096 in one pass: the split-bf16 tiles would drift (1.1e-5); the dispatcher keeps reductions
    longer than 1 024 per workgroup off that pipe"""
    from gaot_amd import ops
    g = torch.Generator().manual_seed(3)
    A, Bm = torch.rand(1024, 4096, generator=g) + 0.5, torch.rand(1024, 4096, generator=g) + 0.5
    out = ops.linear_nt(A.to(dev()), Bm.to(dev()))
    assert rel(out, A.double() @ Bm.double().t()) < 2e-6


# ------------------------------------------------------------------ grouped weight-gradient products
def _wgrad_items(shapes, seed, with_colsum=()):
    g = torch.Generator().manual_seed(seed)
    items, refs = [], []
    for i, (Mo, No, K) in enumerate(shapes):
        dy = torch.randn(K, Mo, generator=g)
        x = torch.randn(K, No, generator=g)
        out = torch.full((Mo, No), float("nan"), device=dev())
        cs = torch.full((Mo,), float("nan"), device=dev()) if i in with_colsum else None
        items.append((dy.to(dev()), Mo, x.to(dev()), No, out, No, cs, Mo, No, K))
        refs.append((dy.double().t() @ x.double(), dy.double().sum(0)))
    return items, refs


def test_wgrad_grouped_matches_float64_and_is_deterministic():
    """gaot_gemm_tn_grouped: many dW = dY^T X (+ db) in one launch; ragged tiles, one and many K slabs, > 24 products (two
    launches), ticket counters back at zero, bitwise repeatable"""
    from gaot_amd import ops
    shapes = [(256, 256, 2048), (2048, 256, 4096), (132, 68, 1024), (64, 64, 8192), (256, 1024, 1024), (768, 256, 3072),
              (36, 260, 1056), (128, 128, 32)] + [(128, 132, 2048)] * 20
    items, refs = _wgrad_items(shapes, 11, with_colsum=(0, 2, 3, 6))
    ops.wgrad_launch(items)
    torch.cuda.synchronize()
    for it, (ref, cref) in zip(items, refs):
        assert rel(it[4], ref) < 2e-6, it[7:]
        if it[6] is not None:
            assert maxrel(it[6], cref) < 5e-6, it[7:]
    assert int(ops._SCRATCH_DEFAULT[("wgrad_counters", dev().index)].abs().sum()) == 0
    first = [it[4].clone() for it in items]
    for it in items:
        it[4].fill_(float("nan"))
    ops.wgrad_launch(items)
    torch.cuda.synchronize()
    assert all(torch.equal(a, it[4]) for a, it in zip(first, items))


def test_wgrad_grouped_same_signed_long_reduction():
    """thousands of same-signed products per output (the bf16 MFMA accumulator does not round to nearest): the grouped launch keeps
    every matrix-pipe accumulation run at <= 1 024 values of k like the single-product path"""
    from gaot_amd import ops
    g = torch.Generator().manual_seed(3)
    K, Mo, No = 8192, 128, 256
    dy = torch.rand(K, Mo, generator=g) + 0.5
    x = torch.rand(K, No, generator=g) + 0.5
    out = torch.empty(Mo, No, device=dev())
    ops.wgrad_launch([(dy.to(dev()), Mo, x.to(dev()), No, out, No, None, Mo, No, K)])
    assert rel(out, dy.double().t() @ x.double()) < 1.5e-6


def test_deferred_wgrad_scope_equals_immediate_products():
    """ops.deferred_wgrad(): products with a destination are queued and computed at scope exit; ineligible ones run at once"""
    from gaot_amd import ops
    g = torch.Generator().manual_seed(9)
    dy, x = torch.randn(4096, 256, generator=g).to(dev()), torch.randn(4096, 512, generator=g).to(dev())
    small_dy, small_x = torch.randn(4096, 6, generator=g).to(dev()), torch.randn(4096, 10, generator=g).to(dev())
    want = ops.matmul_tn(dy, x)
    want_small = ops.matmul_tn(small_dy, small_x)
    out, db = torch.zeros(256, 512, device=dev()), torch.zeros(256, device=dev())
    out_small = torch.zeros(6, 10, device=dev())
    with ops.deferred_wgrad():
        r = ops.matmul_tn(dy, x, out=out, colsum_out=db, final=True)
        assert r is out and len(ops._WGRAD_QUEUE) == 1
        ops.matmul_tn(small_dy, small_x, out=out_small, final=True)          # not groupable (6 x 10): computed immediately
        assert len(ops._WGRAD_QUEUE) == 1 and rel(out_small, want_small) < 1e-6
        tmp = torch.zeros(256, 512, device=dev())
        ops.matmul_tn(dy, x, out=tmp)                            # a destination somebody may read at once (not `final`): immediate
        assert len(ops._WGRAD_QUEUE) == 1 and rel(tmp, want) < 2e-6
    assert not ops._WGRAD_QUEUE
    assert rel(out, want) < 2e-6 and maxrel(db, dy.double().sum(0)) < 5e-6


def test_colsum_grouped_contiguous_and_column_block_outputs():
    """gaot_colsum_grouped through ops.colsum(final=True) inside a deferral scope: many small partial-row matrices, strided inputs
    (column blocks of one workspace), outputs that are contiguous vectors or column blocks of a wider matrix"""
    from gaot_amd import ops
    g = torch.Generator().manual_seed(21)
    ws = torch.randn(200, 12800, generator=g).to(dev())
    wide = torch.zeros(64, 128, device=dev())
    outs, refs = [], []
    with ops.deferred_wgrad():
        for off, width in ((0, 4096), (4096, 256), (12288, 64), (8192, 4096)):
            o = torch.zeros(width, device=dev())
            assert ops.colsum(ws[:, off:off + width], out=o, final=True) is o
            outs.append(o); refs.append(ws[:, off:off + width].double().sum(0))
        blk = wide[:, 64:]
        assert ops.colsum(ws[:, 4352:4352 + 4096], out=blk, final=True) is blk
        small = torch.randn(37, 256, generator=g).to(dev())
        o2 = torch.zeros(256, device=dev())
        ops.colsum(small, out=o2, final=True)
        assert len(ops._COLSUM_QUEUE) == 6 and float(o2.abs().sum()) == 0.0          # nothing computed yet
    for o, r in zip(outs, refs):
        assert maxrel(o, r) < 2e-6
    assert maxrel(wide[:, 64:], ws[:, 4352:4352 + 4096].double().sum(0).view(64, 64)) < 2e-6 and float(wide[:, :64].abs().sum()) == 0.0
    assert maxrel(o2, small.double().sum(0)) < 2e-6


# ------------------------------------------------------------------ pre-split fp16 weight planes
@pytest.mark.parametrize("M,N,K", [(8192, 2048, 256), (8192, 256, 1024), (4096, 768, 256)])
def test_gemm_with_presplit_f16_weight_planes_is_bit_identical(M, N, K):
    """gaot_gemm_desc.b_planes: the two fp16 pieces of a weight matrix (interleaved per 16-wide k group), built once per pass by gaot_split_f16_planes_grouped from the same
    magnitude word the kernel takes its inverse scale from, are exactly what the tile kernel would have formed itself: forward (x W^T)
    and input-gradient (g W) products are BIT-IDENTICAL with and without them -- also for a row block / column block of a registered
    matrix (fused q|k|v weights, the split recovery weight) -- and the planes are really in use (the debug switch changes the kernel)."""
    from gaot_amd import ops, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    x, gy = torch.randn(M, K, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    W = torch.nn.Parameter((torch.randn(N, K, generator=g) * 0.07).cuda())
    ops.begin_pass()
    ops.refresh_weight_amax([W])
    word, pl, ld, stride = ops.weight_operand(W.detach(), True)
    assert pl is not None and ld == 2 * K and stride == 16
    # the planes themselves: h + m reproduces s * w up to its last bit (|e| <= 2^-23 |s w|, zero for three values in four) for every
    # element within 2^-16 of the largest, to 2^-25 absolute below that
    # layout [row][k / 16][piece][k % 16]: the two pieces of a 16-wide k group are one 64-byte segment
    pk = ops._PLANE_CACHE[(W.data_ptr(), N, K, 0)][0].view(torch.float16).view(N, K // 16, 2, 16).permute(2, 0, 1, 3).reshape(2, N, K).double()
    amax = float(word.view(32, 32)[:, 0].max())          # the 32 slots (heads of the lines; floats 1 / 2 of the first line carry the planes' spread L)
    sc = 2.0 ** (13 - math.floor(math.log2(amax)))
    big = W.detach().abs().double() * sc >= 0.25
    err = ((pk[0] + pk[1]) - W.detach().double() * sc).abs() / (W.detach().abs().double() * sc).clamp_min(1e-300)
    assert float(err[big].max()) <= 2.0 ** -23 and float((err[big] == 0).double().mean()) > 0.70
    assert float((pk[0] + pk[1] - W.detach().double() * sc)[~big].abs().max()) <= 2.0 ** -25
    res = {}
    for on in (1, 0):
        old = lib.gaot_debug_set_gemm_planes(on)
        try:
            res[on] = (ops.linear_nt(x, W.detach()), ops.matmul_nn(gy, W.detach()), ops.linear_nt(x, W.detach()[N // 4:N // 2]), ops.matmul_nn(gy[:, :N // 2].contiguous(), W.detach()[:N // 2]),
                       ops.matmul_nn(gy, W.detach()[:, K // 2:]))
        finally:
            lib.gaot_debug_set_gemm_planes(old)
    for a, b in zip(res[1], res[0]):
        assert torch.equal(a, b)
    assert rel(res[1][0], x.double().cpu() @ W.detach().double().cpu().t()) < 6e-7


def test_trainstep_with_f16_weight_planes_equals_without(monkeypatch):
    """three TrainStep updates (eager and hipGraph) with the pre-split planes (default) and without: the same weights bit for bit"""
    import importlib
    from gaot_amd import ops
    from gaot_amd.trainer import TrainStep
    from tests.test_ddp_gpu import _build, _data, _flat
    lat, x, p, t = _data()
    res = {}
    for planes in (True, False):
        monkeypatch.setattr(ops, "_USE_PLANES", planes)
        for graph in (False, True):
            model = _build(seed=5).to(dev()).train()
            ts = TrainStep(model, lr=2e-3, weight_decay=1e-4, use_graph=graph)
            ts.bind(p.to(dev()), t.to(dev()), latent_tokens_coord=lat.to(dev()), xcoord=x.to(dev()))
            for _ in range(3):
                ts.step()
            torch.cuda.synchronize()
            res[(planes, graph)] = _flat(model).cpu()
    assert torch.equal(res[(True, False)], res[(True, True)]) and torch.equal(res[(True, False)], res[(False, False)]) and torch.equal(res[(False, False)], res[(False, True)])


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,K2", [(8192, 256, 256, 256), (4096, 384, 128, 64), (300, 256, 64, 32)])
def test_linear_over_a_concatenated_input_on_the_split_tiles(M, N, K, K2):
    """y = [x | x2] W^T without forming the concatenation (the long skip connections of the processor, attn.py): with fp16 pieces the
    product runs on the split tiles like any other -- columns k >= K come from the second operand, both halves are scaled through
    the LARGER of their two magnitude words.  fp32-level against float64 whatever the two operands' magnitudes, words of x and x2 carried
    into the weight-gradient products, and the same results with three bf16 pieces."""
    from gaot_amd import ops, _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + K2)
    for ma, mb in ((1.0, 1.0), (1e-3, 50.0), (2e4, 1e-2)):
        x, x2 = torch.randn(M, K, generator=g) * ma, torch.randn(M, K2, generator=g) * mb
        w, b = torch.randn(N, K + K2, generator=g) * 0.05, torch.randn(N, generator=g)
        dy = torch.randn(M, N, generator=g)
        xd, x2d, wd, bd = (t.cuda().requires_grad_(True) for t in (x, x2, w, b))
        ops.begin_pass()
        y = ops.linear(xd, wd, bd, x2=x2d)
        on_split = lib.gaot_debug_last_gemm_path() == 3
        y.backward(dy.cuda())
        X = torch.cat([x, x2], 1).double()
        ref = X @ w.double().t() + b.double()
        # the bound of two-piece operands is relative to the LARGER half's magnitude: compare with the product's own scale
        assert rel(y, ref) < 6e-7, (ma, mb, rel(y, ref))
        assert rel(wd.grad, dy.double().t() @ X) < 1e-6
        assert rel(xd.grad, dy.double() @ w.double()[:, :K]) < 6e-7 and rel(x2d.grad, dy.double() @ w.double()[:, K:]) < 6e-7
        if M >= 8192 and N % 128 == 0:
            assert on_split, "the concatenated-input product should be on the split tiles at this size"
        old = ops.set_f32_pieces("bf16x3")
        try:
            y3 = ops.linear(xd.detach(), wd.detach(), bd.detach(), x2=x2d.detach())
        finally:
            ops.set_f32_pieces(old)
        assert rel(y3, ref) < 6e-7 and rel(y, y3) < 6e-7


@pytest.mark.gpu
def test_split_k_products_publish_their_output_word():
    """gaot_gemm_desc.c_absmax with split_k > 1: the reduce launch publishes max |C| (the residual stream a long skip connection
    re-reads is the output of a split-K product: without the word its consumers would each pay an absmax launch)"""
    from gaot_amd import ops
    g = torch.Generator().manual_seed(5)
    x, w, r = torch.randn(8192, 2048, generator=g).cuda(), (torch.randn(256, 2048, generator=g) * 0.03).cuda(), torch.randn(8192, 256, generator=g).cuda()
    ops.begin_pass()
    y = ops.linear_nt(x, w, residual=r, ldr=256)
    word = ops.gemm.last_c_amax
    assert ops._split_for_narrow_output(8192, 256, 2048) > 1 and word is not None          # (K = 1 024 runs without slabs since round 5)
    assert float(word.max()) == float(y.abs().max())
    assert rel(y, x.double().cpu() @ w.double().cpu().t() + r.double().cpu()) < 6e-7


@pytest.mark.gpu
def test_one_launch_mse_loss_and_gradient():
    """ops.mse_loss_and_grad: loss and d loss / d pred for a unit seed from ONE launch -- the same bits as the autograd form
    (mse_loss + backward), the optional optimizer tick advanced by exactly one, the ticket back at zero (repeatable)."""
    from gaot_amd import ops
    g = torch.Generator().manual_seed(3)
    for n in (8 * 16384, 1000003, 7):
        p, y = torch.randn(n, generator=g).cuda().requires_grad_(True), torch.randn(n, generator=g).cuda()
        if n % 4 == 0:
            p2 = p.detach().view(8, -1, 1).requires_grad_(True)
            y2 = y.view(8, -1, 1)
        else:
            p2, y2 = p, y
        l0 = ops.mse_loss(p2, y2)
        l0.backward()
        tick = torch.full((1,), 41.0, device="cuda")
        for _ in range(2):
            l1, dp = ops.mse_loss_and_grad(p2.detach(), y2, tick=tick)
            assert torch.equal(l1, l0.detach()) and torch.equal(dp, p2.grad) and dp.shape == p2.shape
        assert float(tick) == 43.0
        assert rel(l1, ((p.detach().double() - y.double()) ** 2).mean()) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,F", [(8192, 256, 1024), (2048, 512, 1024), (1000, 256, 512), (4096, 384, 1536)])
def test_normed_swiglu_ffn_is_the_composition_of_its_parts(M, K, F):
    """ops.normed_swiglu_ffn (one node: the norm-gradient kernel sums the K slabs of du [w1; w3] itself, gaot_gemm_desc.raw_slabs +
    gaot_rmsnorm_bwd_slabs) against rms_norm followed by swiglu_ffn with the residual on the normalised stream: the same forward
    bits; the same gradients (bit for bit while the slab sums run in the same order, i.e. up to four slabs) and fp32-level against
    float64."""
    from gaot_amd import ops
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(4, M // 4, K, generator=g)
    wn = 1.0 + 0.1 * torch.randn(K, generator=g)
    w1, w3, w2 = (torch.randn(F, K, generator=g) * 0.06, torch.randn(F, K, generator=g) * 0.06, torch.randn(K, F, generator=g) * 0.03)
    dy = torch.randn(4, M // 4, K, generator=g)
    outs = []
    for fused in (True, False):
        xd, wnd, w1d, w3d, w2d = (t.cuda().requires_grad_(True) for t in (x, wn, w1, w3, w2))
        ops.begin_pass()
        if fused:
            y = ops.normed_swiglu_ffn(xd, wnd, 1e-6, w1d, w3d, w2d)
            assert y is not None
        else:
            h = ops.rms_norm(xd, wnd, 1e-6)
            y = ops.swiglu_ffn(h, w1d, w3d, w2d, residual=h)
        y.backward(dy.cuda())
        outs.append([y.detach()] + [t.grad for t in (xd, wnd, w1d, w3d, w2d)])
    assert torch.equal(outs[0][0], outs[1][0])
    nz = ops._split_for_narrow_output(M, K, 2 * F)
    for a, b, name in zip(outs[0][1:], outs[1][1:], ("dx", "dwn", "dw1", "dw3", "dw2")):
        if nz <= 4:
            assert torch.equal(a, b), name
        assert rel(a, b) < 2e-6, (name, rel(a, b))
    # float64
    X, Wn, W1, W3, W2 = (t.double().requires_grad_(True) for t in (x, wn, w1, w3, w2))
    hh = X * torch.rsqrt((X * X).mean(-1, keepdim=True) + 1e-6) * Wn
    yy = hh + (torch.nn.functional.silu(hh @ W1.t()) * (hh @ W3.t())) @ W2.t()
    yy.backward(dy.double())
    assert rel(outs[0][0], yy.detach()) < 6e-7
    for a, r, name in zip(outs[0][1:], (X, Wn, W1, W3, W2), ("dx", "dwn", "dw1", "dw3", "dw2")):
        assert rel(a, r.grad) < 2e-6, (name, rel(a, r.grad))


@pytest.mark.gpu
@pytest.mark.parametrize("Q,C,Cout,OC,bias", [(16384, 64, 64, 1, True), (5000, 64, 128, 3, False), (300, 32, 64, 4, True)])
def test_proj_fold_single_launch_node(Q, C, Cout, OC, bias, monkeypatch):
    """ops.proj_fold (the decoder's output projection folded into the recovery block) as ONE launch each way (gaot_proj_fold_fwd / _bwd:
    one pass over the [Q, C] row bias, the last workgroup finishes the small matrices) against the same node as library products and
    against float64; the recovery weight's column block arrives as a strided view, exactly as the model hands it over."""
    from gaot_amd import ops
    g = torch.Generator().manual_seed(Q + OC)
    hw = (torch.randn(OC, C, generator=g) * 0.3)
    hb = torch.randn(OC, generator=g) if bias else None
    wr = torch.randn(C, Cout + 32, generator=g) * 0.2            # the recovery weight; its first Cout columns are Wr_a
    rowb = torch.randn(Q, C, generator=g)
    gw, gr = torch.randn(OC, Cout, generator=g), torch.randn(Q, OC, generator=g)
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(ops, "_FUSED_PROJ_FOLD", fused)
        hwd, wrd, rbd = (t.cuda().requires_grad_(True) for t in (hw, wr, rowb))
        hbd = hb.cuda().requires_grad_(True) if bias else None
        weff, rproj = ops.proj_fold(hwd, hbd, wrd[:, :Cout], rbd)
        torch.autograd.backward([weff, rproj], [gw.cuda(), gr.cuda()])
        outs.append([weff.detach(), rproj.detach(), hwd.grad, wrd.grad, rbd.grad] + ([hbd.grad] if bias else []))
    HW, WR, RB = (t.double().requires_grad_(True) for t in (hw, wr, rowb))
    HB = hb.double().requires_grad_(True) if bias else None
    weff64 = HW @ WR[:, :Cout]
    rproj64 = RB @ HW.t() + (HB if bias else 0.0)
    torch.autograd.backward([weff64, rproj64], [gw.double(), gr.double()])
    refs = [weff64.detach(), rproj64.detach(), HW.grad, WR.grad, RB.grad] + ([HB.grad] if bias else [])
    for a, b, r, name in zip(outs[0], outs[1], refs, ("weff", "rproj", "dW", "dWr", "drowb", "db")):
        assert a.shape == r.shape, name
        assert rel(a, r) < 2e-6 and rel(b, r) < 2e-6, (name, rel(a, r), rel(b, r))
    # deterministic
    monkeypatch.setattr(ops, "_FUSED_PROJ_FOLD", True)
    hwd, wrd, rbd = (t.cuda().requires_grad_(True) for t in (hw, wr, rowb))
    weff, rproj = ops.proj_fold(hwd, None, wrd[:, :Cout], rbd)
    torch.autograd.backward([weff, rproj], [gw.cuda(), gr.cuda()])
    again = [hwd.grad.clone(), wrd.grad.clone()]
    hwd.grad = wrd.grad = rbd.grad = None
    weff, rproj = ops.proj_fold(hwd, None, wrd[:, :Cout], rbd)
    torch.autograd.backward([weff, rproj], [gw.cuda(), gr.cuda()])
    assert torch.equal(again[0], hwd.grad) and torch.equal(again[1], wrd.grad)
