#!/usr/bin/env python3
"""bench.py -- GAOT training throughput on MI355X (contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one reference-trainer step (forward + MSE + backward + AdamW, plus the flat-gradient RCCL all-reduce
when N > 1) on BASELINE config 2: 2-D mesh with 16 384 nodes, batch 8 PER GPU (weak scaling), the reference's
example model (config/examples/time_indep/poisson_gauss.json: latent 64x64, C=64, patch 2, transformer 256 x 3
blocks, 8 heads; 3 396 033 parameters), synthetic data, random-init weights, inputs resident in HBM.
The HEADLINE (`value`, `ms_per_step`, `dtype: "f32"`) is fp32-equivalent arithmetic end to end, like the reference (no autocast
anywhere, base_trainer.py:63-68): storage and accumulation are fp32, and every product on the matrix pipe carries each fp32 operand
at fp32 width -- as TWO fp16 pieces of the operand scaled by a power of two read from a device-resident magnitude word (24
significant bits up to the operand's last one: |s x - h - m| <= 2^-23 |s x|, zero for three values in four; three piece products), or,
where no word is available, as THREE bf16 pieces (six
piece products).  Its error against the float64 oracle is in `rel_l2_vs_oracle.vs_float64_oracle` next to the fp32 reference's own.
Other arithmetic is reported as labelled `variants` (bf16x3: the three-piece products everywhere, equally fp32-level; bf16x2: two
rounded bf16 pieces, 16 bits; bf16: one piece in the tile GEMMs), each with its own error against the float64 oracle.

Extra objects on the JSON line:
  roofline     dominant kernel family = the split-bf16 MFMA GEMM kernels behind gaot_gemm_f32 and the grouped weight-gradient
               launch: piece-product FLOPs ISSUED on the bf16 matrix pipe in one step / their summed duration, timed live with HIP
               events on the launch stream in an instrumented eager step; peak = 2500 TFLOP/s (dense bf16 MFMA, MI355X_MICROARCH.md);
               the fp32-equivalent rate (algorithmic FLOPs) is a secondary key
  cpu_baseline the CPU oracle (oracle/gaot_oracle.py, a parity-checked port of the reference path) timed on the
               host cores of this box on the same workload, a few steps (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace as NS

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_NODES, BATCH, LATENT, C_LIFT, HIDDEN, PATCH, RADIUS = 16384, 8, [64, 64], 64, 256, 2, 0.033
PEAK_F32_MATRIX_TFLOPS = 157.3
PEAK_BF16_MATRIX_TFLOPS = 2500.0
PEAK_HBM_GBPS = 8000.0


def build_model():
    from gaot_amd.model.gaot import GAOT
    from gaot_amd.model.layers.magno import MAGNOConfig
    from gaot_amd.model.layers.attn import TransformerConfig
    mcfg = MAGNOConfig(coord_dim=2, radius=RADIUS, hidden_size=64, mlp_layers=3, lifting_channels=C_LIFT)
    tcfg = TransformerConfig(patch_size=PATCH, hidden_size=HIDDEN)
    return GAOT(1, 1, NS(args=NS(magno=mcfg, transformer=tcfg), latent_tokens_size=LATENT))


def synthetic(seed: int, device):
    g = torch.Generator().manual_seed(seed)
    ax = torch.linspace(-1, 1, LATENT[0])
    lat = torch.stack(torch.meshgrid(ax, torch.linspace(-1, 1, LATENT[1]), indexing="ij"), -1).reshape(-1, 2)
    x = torch.rand(N_NODES, 2, generator=torch.Generator().manual_seed(0)) * 2 - 1      # same mesh on every rank
    p = torch.randn(BATCH, N_NODES, 1, generator=g)
    t = torch.randn(BATCH, N_NODES, 1, generator=g)
    return lat.to(device), x.to(device), p.to(device), t.to(device)


GNO_CALLS = {"gaot_gno_lift_gather_reduce": "encoder fwd", "gaot_gno_lift_gather_reduce_ep": "encoder fwd",
             "gaot_gno_lift_edge_grad": "encoder bwd",
             "gaot_gno_proj_gather_reduce": "decoder fwd", "gaot_gno_proj_gather_reduce_bin": "decoder fwd",
             "gaot_gno_proj_backward": "decoder bwd", "gaot_gno_proj_gather_t_ep": "decoder bwd", "gaot_gno_proj_gather_t_ep_w": "decoder bwd",
             "gaot_gno_gather_reduce": "unfused transform", "gaot_gno_edge_grad": "unfused edge grad"}


def _instrument_gno(lib):
    """HIP events (launch stream) around every fused integral-transform entry point; returns (records, saved originals)"""
    gno, saved = [], {}
    for name in GNO_CALLS:
        fn = getattr(lib, name)
        saved[name] = fn

        def wrap(fn=fn, name=name):
            def call(*a):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                rc = fn(*a)
                e.record()
                gno.append((name, s, e))
                return rc
            return call
        setattr(lib, name, wrap())
    return gno, saved


def _gno_us(gno):
    out = {}
    for name, s_, e_ in gno:
        out[GNO_CALLS[name]] = out.get(GNO_CALLS[name], 0.0) + 1e3 * s_.elapsed_time(e_)
    return out


def gno_algorithmic_bytes(Ee, Ed, Q, n_phys, C, B, cin, cout):
    """every operand of the four fused integral-transform launches touched once (k_e rows, node features, CSR arrays, outputs);
    B = batch inside the launch (vx unions: 1, with the union's node counts)"""
    return {
        "encoder fwd": 4.0 * (Ee * C + B * n_phys * cin + 3 * Ee + Q + B * Q * C),
        "encoder bwd": 4.0 * (B * Q * C + Ee * C + B * n_phys * cin + 3 * Ee + Ee * C),
        "decoder fwd": 4.0 * (Ed * C + B * Q * C + 3 * Ed + n_phys + B * n_phys * cout),
        "decoder bwd": 4.0 * (B * n_phys * cout + Ed * C + B * Q * C + 5 * Ed + Ed * C + B * Q * C),
    }


def gemm_roofline(ts):
    """Instrumented EAGER step: HIP events (torch.cuda.Event on the launch stream = torch's current stream) around
    every gaot_gemm_f32 call; returns achieved TFLOP/s of the GEMM family over one step."""
    from gaot_amd import ops, _lib
    lib = _lib.load()
    records = []
    raw = ops.gemm

    def timed_gemm(M, N, K, *a, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = raw(M, N, K, *a, **kw)
        e.record()
        kind = "tn" if not int(a[2]) else ("nt" if int(a[5]) else "nn")
        pcs = kw.get("pieces") or ops._PIECES[kind]
        records.append((s, e, 2.0 * M * N * K, (M, N, K, int(a[2]), int(a[5]), kw.get("split_k", 1)), lib.gaot_debug_last_gemm_path(), pcs))
        return out

    # the HBM regime: the fused gather / segment-reduce / edge-gradient kernels of the integral transforms (csrc/gno.hip)
    gno, saved = _instrument_gno(lib)
    wg = []
    raw_wgrad = ops.wgrad_launch

    def timed_wgrad(items):           # the grouped weight-gradient launch (every dW product of the backward pass in one kernel)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        raw_wgrad(items)
        e.record()
        wg.append((s, e, sum(2.0 * it[7] * it[8] * it[9] for it in items), sum(4.0 * (it[9] * (it[7] + it[8]) + it[7] * it[8]) for it in items), len(items),
                   ops._PIECES["tn"]))

    use_graph = ts.use_graph
    ts.use_graph = False
    ts.step()                      # untimed eager step (allocator warm)
    gno.clear()
    ops.gemm = timed_gemm
    ops.wgrad_launch = timed_wgrad
    try:
        torch.cuda.synchronize()
        # park the GPU behind a ~60 ms spin so the host enqueues the whole eager step ahead of it: the events then
        # bracket back-to-back GPU execution instead of host launch latency
        torch.cuda._sleep(int(60e-3 * 2.0e9))
        ts.step()
        torch.cuda.synchronize()
    finally:
        ops.gemm = raw
        ops.wgrad_launch = raw_wgrad
        ts.use_graph = use_graph
        for name, fn in saved.items():
            setattr(lib, name, fn)
    gno_us = _gno_us(gno)
    # piece products per fp32-equivalent product: three bf16 pieces -> 6; two bf16 pieces -> 3; the default "f32" precision on the split
    # tiles = two fp16 pieces of the scaled operand -> 3 (ops._F16_PIECES; 6 with GAOT_F32_PIECES=bf16x3)
    nprod = {3: 3.0 if ops._F16_PIECES[0] else 6.0, 2: 3.0, 1: 1.0}
    one_piece = os.environ.get("GAOT_BENCH_ONE_PIECE") == "1"          # the `bf16` variant (gaot_debug_set_gemm_pieces(1)): one product
    mfma = [r for r in records if r[4] == 3]      # launches served by the split-bf16 MFMA tile kernels (the bf16 matrix pipe)
    f32m = [r for r in records if r[4] == 1]      # fp32-MFMA tile launches (products too small / narrow for the split tiles)
    n_split = len(mfma)
    ms = sum(r[0].elapsed_time(r[1]) for r in mfma) + sum(w[0].elapsed_time(w[1]) for w in wg)
    flops = sum(r[2] for r in mfma) + sum(w[2] for w in wg)
    piece_flops = sum(r[2] * (1.0 if one_piece else nprod[r[5]]) for r in mfma) + sum(w[2] * (1.0 if one_piece else nprod[w[5]]) for w in wg)
    abytes = sum(4.0 * (r[3][0] * r[3][2] + r[3][1] * r[3][2] + r[3][0] * r[3][1]) for r in mfma) + sum(w[3] for w in wg)   # A + B + C touched once
    ms_all = sum(r[0].elapsed_time(r[1]) for r in records) + sum(w[0].elapsed_time(w[1]) for w in wg)
    if os.environ.get("GAOT_BENCH_GEMM_TABLE"):
        rows = sorted(((r[0].elapsed_time(r[1]) * 1e3, r[2], r[3]) for r in records), key=lambda t: -t[0])
        for us, fl, (M, N, K, ak, bk, sk) in rows:
            print(f"# gemm M={M:6d} N={N:5d} K={K:6d} a_k={ak} b_k={bk} split={sk:3d} {us:8.1f}us {fl / us / 1e6:6.1f}TF", file=sys.stderr)
    if os.environ.get("GAOT_BENCH_GEMM_TABLE"):
        for w in wg:
            us = w[0].elapsed_time(w[1]) * 1e3
            print(f"# grouped weight gradients: {w[4]} products in one launch {us:8.1f}us {w[2] / us / 1e6:6.1f}TF", file=sys.stderr)
    return {"launches": len(mfma) + len(wg), "flops": flops, "ms": ms, "tflops": flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
            "piece_tflops": piece_flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, "piece_flops": piece_flops,
            "f32_mfma_launches": len(f32m), "f32_mfma_ms": sum(r[0].elapsed_time(r[1]) for r in f32m), "f32_mfma_gflop": sum(r[2] for r in f32m) / 1e9,
            "skinny_launches": len(records) - len(mfma) - len(f32m), "all_gemm_ms": ms_all, "split_launches": n_split,
            "alg_bytes_per_launch": abytes / max(1, len(mfma) + len(wg)), "gno_us": gno_us, "gno_launches": len(gno),
            "grouped_wgrad": {"launches": len(wg), "products": sum(w[4] for w in wg), "gflop": sum(w[2] for w in wg) / 1e9,
                              "us": sum(w[0].elapsed_time(w[1]) for w in wg) * 1e3}}


def recorded_traffic(family: str = "gemm"):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE, see
    tools/pmc_traffic.py); bench.py cannot run the profiler on itself.  profiles/current_traffic.json names, per family, the
    file the number comes from and the commit it was measured at, so a stale figure is visible as such."""
    try:
        with open(os.path.join(ROOT, "profiles", "current_traffic.json")) as f:
            cur = json.load(f)[family]
        with open(os.path.join(ROOT, "profiles", cur["file"])) as f:
            return json.load(f)["hbm_bytes_per_launch"], f"profiles/{cur['file']} (measured at commit {cur['commit']})"
    except Exception:
        return None, None


def cpu_baseline(sd, tensors, hip, steps: int = 3):
    """the CPU oracle on the same weights and the same batch: timed as the baseline, and its FIRST step (same initial
    weights as `hip` = prediction / loss / gradients of the HIP path before any update) is the reference of the
    rel-L2 half of BASELINE's metric."""
    from oracle import gaot_oracle as O
    lat, x, p, t = [v.cpu() for v in tensors]
    cfg = O.OracleConfig(radius=RADIUS, hidden_size=64, lifting_channels=C_LIFT, patch_size=PATCH, tf_hidden_size=HIDDEN,
                         latent_tokens_size=LATENT, precompute_edges=True)
    # explicit-difference distance test (the reference's `grid` backend = method 'auto'); the HIP cell list must have built
    # exactly this graph (checked), so both sides integrate over the same edges
    enc, dec = [O.radius_csr(x, lat, RADIUS, exact=True)], [O.radius_csr(lat, x, RADIUS, exact=True)]
    same_graph = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in ((enc[0], hip["enc_csr"]), (dec[0], hip["dec_csr"])))
    batch = dict(latent=lat, xcoord=x, pndata=p, target=t, encoder_nbrs=enc, decoder_nbrs=dec)
    # the same step with the geometry statistics evaluated in float64 (oracle test instrument): four gradient tensors sit right
    # behind ReLU gates on those statistics and move by ~2e-4 when the ORACLE ITSELF switches their arithmetic
    cfg64 = O.OracleConfig(**{**cfg.__dict__, "stats_dtype": "float64"})
    _, grads64, _, _ = O.train_step(sd, cfg64, batch, state=None)
    dbl = lambda v: v.double() if v.is_floating_point() else v
    batch64 = {k: (dbl(v) if torch.is_tensor(v) else v) for k, v in batch.items()}
    _, gradsd, _, _, predd = O.train_step({k: dbl(v) for k, v in sd.items()}, cfg, batch64, state=None, return_pred=True)   # float64 end to end
    loss0, grads0, sd, state, pred0 = O.train_step(sd, cfg, batch, state=None, return_pred=True)      # warm-up + parity reference
    top = max(float(g.double().norm()) for g in grads0.values())
    rel = lambda ref: {k: float((hip["grads"][k].double() - g.double()).norm()) / max(float(g.double().norm()), 1e-3 * top) for k, g in ref.items()}
    gerr, gerr64 = rel(grads0), rel(grads64)
    worst, worst64 = max(gerr, key=gerr.get), max(gerr64, key=gerr64.get)
    # and against the SAME oracle evaluated in float64 throughout (weights, batch, every intermediate): the fp32 oracle's own rounding
    # leaves the comparison, what remains is the kernels' (tools/grad_errors.py prints the per-tensor table)
    topd = max(float(g.norm()) for g in gradsd.values())
    reld = lambda a, b: float((a.double() - b).norm()) / max(float(b.norm()), 1e-3 * topd)
    gerrd = {k: reld(hip["grads"][k], g) for k, g in gradsd.items()}
    gerrd_oracle = {k: reld(grads0[k], g) for k, g in gradsd.items()}
    worstd = max(gerrd, key=gerrd.get)
    vs64 = {"output": float((hip["pred"].double() - predd).norm() / predd.norm()),
            "grad_worst_tensor": gerrd[worstd], "grad_worst_name": worstd,
            "fp32_oracle_output": float((pred0.double() - predd).norm() / predd.norm()),
            "fp32_oracle_grad_worst_tensor": max(gerrd_oracle.values()),
            "what": "the same comparison against the oracle evaluated in float64 end to end; fp32_oracle_*: the fp32 oracle "
                    "(= the reference's arithmetic) against it, i.e. the reference's own rounding on this batch"}
    parity = {"output": float((hip["pred"].double() - pred0.double()).norm() / pred0.double().norm()),
              "vs_float64_oracle": vs64,
              "loss": abs(hip["loss"] - float(loss0)) / abs(float(loss0)),
              "grad_worst_tensor": gerr[worst], "grad_worst_name": worst,
              "grad_worst_tensor_vs_oracle_with_f64_statistics": gerr64[worst64], "grad_worst_name_f64_statistics": worst64,
              "radius_graph_identical": bool(same_graph),
              "what": "HIP path vs CPU oracle, same initial weights and batch as the timed run (rank 0): relative L2 of the "
                      "[8,16384,1] prediction, relative loss error, worst per-tensor relative L2 over the 72 gradient tensors"}
    # the oracle is plain torch ops: oversubscribing a big host slows it down, so take the best of a few thread counts
    all_threads = torch.get_num_threads()
    best, best_dt = all_threads, None
    for nt in sorted({min(all_threads, c) for c in (16, 32, 64, all_threads)}):
        torch.set_num_threads(nt)
        t0 = time.perf_counter()
        _, _, sd, state = O.train_step(sd, cfg, batch, state=state)
        dt1 = time.perf_counter() - t0
        if best_dt is None or dt1 < best_dt:
            best, best_dt = nt, dt1
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    for _ in range(steps):
        _, _, sd, state = O.train_step(sd, cfg, batch, state=state)
    dt = time.perf_counter() - t0
    base = {"value": BATCH * steps / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} full train steps (fwd+MSE+bwd+AdamW) of the same workload (16384 nodes, batch {BATCH}) "
                      f"on the CPU oracle, {dt / steps:.2f} s/step"}
    return base, parity, {"pred": predd, "grads": gradsd}


def _to_dev(o, dev):
    if torch.is_tensor(o):
        return o.to(dev)
    if isinstance(o, (list, tuple)):
        return type(o)(_to_dev(v, dev) for v in o)
    if isinstance(o, dict):
        return {k: _to_dev(v, dev) for k, v in o.items()}
    return o


def torch_gpu_steps(sd, ocfg, okw, target, dev, B, steps: int = 8, warmup: int = 3):
    """`steps` training steps of the reference's algorithm as plain eager PyTorch-ROCm ops on `dev` (see torch_gpu_baseline), driven the way
    the reference trainer drives it (optimizers.py:247-257); returns (samples/s, ms per step, first prediction, first loss)"""
    import dataclasses
    from oracle import gaot_oracle as O
    cfg = dataclasses.replace(ocfg, library_attention=True)
    kw = _to_dev(okw, dev)
    tgt = target.to(dev)
    params = {k: torch.nn.Parameter(v.to(dev).clone(), requires_grad=not k.endswith("rotary_emb.freqs")) for k, v in sd.items()}
    opt = torch.optim.AdamW([q for q in params.values() if q.requires_grad], lr=8e-4, weight_decay=1e-5)

    def one():
        opt.zero_grad()
        pred = O.gaot_forward(params, cfg, **kw)
        loss = torch.mean((pred - tgt) ** 2)
        loss.backward()
        opt.step()
        return pred.detach(), loss.detach()

    with torch.device(dev):          # the restatement creates its scratch tensors on the default device
        pred0, loss0 = one()
        for _ in range(warmup - 1):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    return B / dt, 1e3 * dt, pred0, float(loss0)


def torch_gpu_baseline(sd, tensors, hip, dev, steps: int = 8, warmup: int = 3):
    """The SAME-NODE denominator of the north_star's ">= 5x the reference's single-GPU PyTorch": the reference's algorithm in plain
    eager PyTorch-ROCm ops on this GPU -- the parity-checked restatement (oracle/gaot_oracle.py: matmuls through rocBLAS / hipBLASLt,
    F.scaled_dot_product_attention as attn.py:114 calls it, ATen index_add_ / scatter_reduce where the reference calls torch_scatter, which
    this image does not have) driven the way the reference trainer drives it (optimizers.py:247-257: zero_grad, forward, MSE, backward,
    torch.optim.AdamW), same weights, same batch, same neighbour lists, fp32.  A reported baseline, not the product: nothing here is
    on the HIP path, and the HIP path never touches it."""
    from oracle import gaot_oracle as O
    lat, x, p, t = tensors
    cfg = O.OracleConfig(radius=RADIUS, hidden_size=64, lifting_channels=C_LIFT, patch_size=PATCH, tf_hidden_size=HIDDEN,
                         latent_tokens_size=LATENT, precompute_edges=True)
    rate, ms, pred0, loss0 = torch_gpu_steps(sd, cfg, dict(latent=lat, xcoord=x, pndata=p, encoder_nbrs=[hip["enc_csr"]], decoder_nbrs=[hip["dec_csr"]]),
                                             t, dev, BATCH, steps, warmup)
    ref = hip["pred"].to(dev).double()
    out = {"output_vs_hip": float((pred0.double() - ref).norm() / ref.norm()), "loss_vs_hip": abs(loss0 - hip["loss"]) / abs(hip["loss"])}
    return {"value": rate, "unit": "samples/s", "ms_per_step": ms, "steps": steps, "kind": "port",
            "what": "the reference's algorithm as plain eager PyTorch-ROCm ops on the same MI355X (oracle/gaot_oracle.py on cuda:0 with "
                    "F.scaled_dot_product_attention and torch.optim.AdamW as the reference calls them; ATen index_add_ / scatter_reduce "
                    "stand in for torch_scatter, absent from the image), same weights / batch / neighbour lists, fp32, fwd + MSE + bwd + AdamW",
            "first_step_vs_hip": {"output_rel_l2": out["output_vs_hip"], "loss_rel": out["loss_vs_hip"]}}


def errors_vs_float64(hip, ref64):
    """output rel-L2 and worst per-tensor gradient rel-L2 (denominator floored at 1e-3 of the largest gradient norm) of a HIP
    reference pass against the oracle evaluated in float64 end to end"""
    topd = max(float(g.norm()) for g in ref64["grads"].values())
    errs = {k: float((hip["grads"][k].double() - g).norm()) / max(float(g.norm()), 1e-3 * topd) for k, g in ref64["grads"].items()}
    worst = max(errs, key=errs.get)
    return {"output": float((hip["pred"].double() - ref64["pred"]).norm() / ref64["pred"].norm()),
            "grad_worst_tensor": errs[worst], "grad_worst_name": worst,
            "what": "vs the CPU oracle evaluated in float64 end to end, same initial weights and batch; gradients through the training path"}


def reference_loop_rate(dev, steps: int = 30, warmup: int = 8):
    """The SAME workload driven the way the reference trainer drives it (optimizers.py:247-257 around
    static_trainer.py:160-178): per step, upload the batch AND the coordinates from host memory (new device tensors every
    step), optimizer.zero_grad(), eager forward, nn.MSELoss, eager backward, torch.optim.AdamW.step(); no hipGraph, no flat
    buffers, nothing bound ahead of time.  The geometry caches are hit through the device-side content guard (plan.py)."""
    from gaot_amd import ops
    ops.register_grad_slots([], [])               # the headline run's flat gradient bucket is not part of this loop
    torch.manual_seed(0)
    model = build_model().to(dev).train()
    lat, x, p, t = synthetic(1234, torch.device("cpu"))
    opt = torch.optim.AdamW(model.parameters(), lr=8e-4, weight_decay=1e-5)
    loss_fn = torch.nn.MSELoss()

    def one():
        xb, yb = p.to(dev), t.to(dev)
        latd, coord = lat.to(dev), x.to(dev)
        opt.zero_grad()
        loss = loss_fn(model(latent_tokens_coord=latd, xcoord=coord, pndata=xb), yb)
        loss.backward()
        opt.step()
        return loss.detach()

    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": BATCH * steps / dt, "unit": "samples/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
            "what": "reference-shaped loop: per-step host->device upload of batch and coordinates (pageable memory), "
                    "zero_grad, eager forward/backward, nn.MSELoss, torch.optim.AdamW -- the drop-in speed a user of the "
                    "reference trainer sees without changing it"}


def variant_rate(dev, name: str, ref64, steps: int = 40, warmup: int = 8):
    """The SAME TrainStep in another arithmetic -- a labelled variant, never `value`:
      bf16x3: fp32-level like the headline, by three bf16 pieces per operand and six piece products instead of two fp16 pieces and three
      bf16x2: every product on the bf16 matrix pipe takes each fp32 operand as TWO rounded bf16 pieces (16 significant bits; GEMM tiles,
              grouped weight gradients, attention, the GELU kernel MLP); storage / accumulation fp32
      bf16:   BASELINE configs[1]'s "bf16": tile-GEMM operands rounded to ONE bf16 piece, fp32 accumulation; everything else as the headline
    with its own error against the float64 oracle (same initial weights and batch as the headline)."""
    from gaot_amd import ops, _lib
    from gaot_amd.trainer import TrainStep
    lib = _lib.load()
    old = ops.set_precision("bf16x2") if name == "bf16x2" else dict(ops._PIECES)
    old1 = lib.gaot_debug_set_gemm_pieces(1) if name == "bf16" else None
    old3 = ops.set_f32_pieces("bf16x3") if name == "bf16x3" else None
    try:
        ops.register_grad_slots([], [])
        torch.manual_seed(0)
        model = build_model().to(dev).train()
        lat, x, p, t = synthetic(1234, dev)
        ts = TrainStep(model, lr=8e-4, weight_decay=1e-5, use_graph=True)
        ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
        err = errors_vs_float64(hip_reference_pass(ts, model, (lat, x, p, t)), ref64) if ref64 is not None else None
        for _ in range(warmup):
            ts.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ts.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        del ts
    finally:
        ops.set_gemm_pieces(**old)
        if old1 is not None:
            lib.gaot_debug_set_gemm_pieces(old1)
        if old3 is not None:
            ops.set_f32_pieces(old3)
    dtype = {"bf16x3": "f32 through THREE bf16 pieces per operand, six piece products (rounds 1-3's exact mode; GAOT_F32_PIECES=bf16x3): the same "
                       "fp32-level arithmetic as the headline by other means, for comparison of speed and of error",
             "bf16x2": "bf16x2 products (two rounded bf16 pieces per f32 operand), f32 accumulate and storage",
             "bf16": "bf16 tile-GEMM operands (one piece, RNE), f32 accumulate and storage; attention / kernel MLP / transforms as the headline"}[name]
    return {"value": BATCH * steps / dt, "unit": "samples/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "dtype": dtype,
            "rel_l2_vs_oracle": err}


def secondary_configs(dev, which=("C3", "C4", "C5"), steps: int = 20, warmup: int = 5, oracle: bool = True):
    """BASELINE configs[2..4] at their NAMED sizes (parity-test shapes, not the headline): per config the hipGraph train step
    (samples/s, ms/step), the HBM-regime roofline of its fused integral-transform launches (HIP events in one instrumented eager
    step, operands touched once / 8 TB/s) and the relative L2 of the HIP forward against the CPU oracle's forward at the initial
    weights (the oracle stays test infrastructure: it is only the checker here, after the timed region)."""
    import numpy as np
    from gaot_amd import _lib
    from gaot_amd.model.gaot import GAOT
    from gaot_amd.model.layers.magno import MAGNOConfig
    from gaot_amd.model.layers.attn import TransformerConfig, AttentionConfig
    from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch
    from gaot_amd.trainer import TrainStep
    from tests._workloads import grid, naca_points, shell_points
    lib = _lib.load()
    out = {}

    def measure(name, model, p, t, kw, ocfg, okw, B, note, dims):
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        with torch.no_grad():
            model.eval()
            y0 = model(pndata=p, **kw).cpu()
            model.train()
        ts = TrainStep(model, lr=8e-4, weight_decay=1e-5, use_graph=True)
        ts.bind(p, t, **kw)
        for _ in range(warmup):
            ts.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            ts.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        # one instrumented eager step for the integral-transform launches
        gno, saved = _instrument_gno(lib)
        ts.use_graph = False
        try:
            ts.step()
            gno.clear()
            torch.cuda.synchronize()
            torch.cuda._sleep(int(30e-3 * 2.0e9))
            ts.step()
            torch.cuda.synchronize()
        finally:
            ts.use_graph = True
            for nm, fn in saved.items():
                setattr(lib, nm, fn)
        us = _gno_us(gno)
        by = gno_algorithmic_bytes(*dims)
        tot_us = sum(v for k, v in us.items() if k in by)
        tot_b = sum(by.values())
        gbps = tot_b / (tot_us * 1e-6) / 1e9 if tot_us > 0 else 0.0
        row = {"workload": note, "samples_per_s": B / dt, "ms_per_step": 1e3 * dt, "steps": steps, "hipgraph": True,
               "roofline_hbm": {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                                "algorithmic_bytes_per_step": tot_b, "kernel_us_per_step": tot_us,
                                "per_kernel_us": {k: v for k, v in us.items()}}}
        if oracle:
            from oracle import gaot_oracle as O
            with torch.no_grad():
                ref = O.gaot_forward(sd, ocfg, **okw)
            row["rel_l2_vs_oracle"] = {"output": float((y0.double() - ref.double()).norm() / ref.double().norm()),
                                       "what": "HIP forward vs the CPU oracle's forward, same initial weights and batch"}
            try:          # the same-node PyTorch baseline of this configuration (torch_gpu_baseline's leg, 5 steps)
                rate, ms, _, _ = torch_gpu_steps(sd, ocfg, okw, t, dev, B, steps=5, warmup=2)
                row["torch_gpu_baseline"] = {"value": rate, "unit": "samples/s", "ms_per_step": ms, "steps": 5, "kind": "port",
                                             "this_config_over_it": row["samples_per_s"] / rate,
                                             "what": "the reference's algorithm as plain eager PyTorch-ROCm ops on the same GPU (see the line's torch_gpu_baseline)"}
            except Exception as e:
                row["torch_gpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()
        out[name] = row
        return ts

    if "C3" in which:
        torch.manual_seed(0)
        B, N, NS_DATA = 16, 8192, 48
        mc = MAGNOConfig(radius=RADIUS, lifting_channels=C_LIFT, precompute_edges=True)
        mk = lambda: GAOT(3, 1, NS(args=NS(magno=mc, transformer=TransformerConfig(patch_size=PATCH, hidden_size=HIDDEN)), latent_tokens_size=LATENT)).to(dev).train()
        model = mk()
        g = torch.Generator().manual_seed(0)
        lat = grid(LATENT)
        # a DATASET of 48 airfoil-shaped meshes (the first 16 are the batch of rounds 2-5; the others spread the point density a little, as a
        # family of geometries does, so the batches' edge totals fall into several capacity buckets), fields and graphs resident on the device
        x_all = torch.stack([naca_points(N, g, 0.15 if i < B else 0.12 + 0.06 * (i % 8) / 7) for i in range(NS_DATA)])
        p_all, t_all = torch.randn(NS_DATA, N, 3, generator=g), torch.randn(NS_DATA, N, 1, generator=g)
        x, p, t = x_all[:B], p_all[:B], t_all[:B]
        ns = NeighborSearch("auto")
        xd_all, latd = x_all.to(dev), lat.to(dev)
        pd_all, td_all = p_all.to(dev), t_all.to(dev)
        enc_all = [[ns(xd_all[i], latd, RADIUS)] for i in range(NS_DATA)]
        dec_all = [[ns(latd, xd_all[i], RADIUS)] for i in range(NS_DATA)]
        enc, dec, xd = enc_all[:B], dec_all[:B], xd_all[:B].contiguous()
        csr = lambda rows: [[(d[0]["neighbors_index"].cpu(), d[0]["neighbors_row_splits"].cpu())] for d in rows]
        deg = torch.cat([e[0]["neighbors_row_splits"][1:] - e[0]["neighbors_row_splits"][:-1] for e in enc])
        e_each = [int(e[0]["neighbors_index"].numel()) for e in enc_all]
        Ee, Ed = sum(e_each[:B]), sum(int(d[0]["neighbors_index"].numel()) for d in dec)
        from oracle import gaot_oracle as O
        ocfg = O.OracleConfig(radius=RADIUS, hidden_size=64, lifting_channels=C_LIFT, patch_size=PATCH, tf_hidden_size=HIDDEN,
                              latent_tokens_size=LATENT, precompute_edges=True)
        Q = LATENT[0] * LATENT[1]
        ts = measure("C3", model, p.to(dev), t.to(dev), dict(latent_tokens_coord=latd, xcoord=xd, encoder_nbrs=enc, decoder_nbrs=dec), ocfg,
                     dict(latent=lat, xcoord=x, pndata=p, encoder_nbrs=csr(enc), decoder_nbrs=csr(dec)), B,
                     f"BASELINE configs[2]: NACA0012-shaped 2D meshes (density ~ exp(-dist to the contour / 0.15)), vx mode, {N} nodes, batch {B}, "
                     f"3 input channels; {Ee} encoder edges over the batch, encoder degree max {int(deg.max())}, {int((deg == 0).sum())} empty latent rows",
                     (Ee, Ed, B * Q, B * N, C_LIFT, 1, 3, 1))
        # ---- the rate that counts for a vx dataset: EVERY STEP ANOTHER BATCH COMPOSITION (the reference's shuffling loader, data_utils.py:272-294
        # with static_trainer.py:180-202).  The captured step re-composes the batch's unions on the device from a table uploaded per step
        # (plan.StaticUnion); a new capture happens only when a batch's edge total falls into a capacity bucket not seen before.
        gsh = torch.Generator().manual_seed(7)
        draw = lambda: torch.randperm(NS_DATA, generator=gsh)[:B].tolist()
        pick = lambda rows, b: [rows[i] for i in b]

        def ts_step(b):
            ix = torch.tensor(b, device=dev)
            return ts.step(pd_all[ix], td_all[ix], xcoord=xd_all[ix], encoder_nbrs=pick(enc_all, b), decoder_nbrs=pick(dec_all, b))
        for _ in range(3 * warmup + 9):
            ts_step(draw())
        torch.cuda.synchronize()
        sets0 = len(ts._graph_sets)
        comps = [draw() for _ in range(max(steps, 8))]
        t0 = time.perf_counter()
        for b in comps:
            ts_step(b)
        torch.cuda.synchronize()
        dsh = (time.perf_counter() - t0) / len(comps)
        fixed = {"samples_per_s": out["C3"]["samples_per_s"], "ms_per_step": out["C3"]["ms_per_step"], "what": "ONE batch bound once and replayed (rounds 2-5's C3 line)"}
        totals = [sum(e_each[i] for i in b) for b in comps]
        out["C3"].update({"samples_per_s": B / dsh, "ms_per_step": 1e3 * dsh, "steps": len(comps), "fixed_batch": fixed,
                          "shuffled": {"distinct_compositions": len({tuple(b) for b in comps}), "dataset_samples": NS_DATA,
                                       "encoder_edges_per_batch": [min(totals), max(totals)],
                                       "captured_steps_before_timing": sets0, "captures_inside_timed_region": len(ts._graph_sets) - sets0,
                                       "edge_buckets": sorted({u.e_cap for v in ts._graph_sets.values() for u in v["unions"][0]}),
                                       "what": "TrainStep.step(pndata, target, xcoord=, encoder_nbrs=, decoder_nbrs=) with a NEW composition of 16 of the "
                                               "dataset's 48 resident meshes every step: hipGraph replay of the whole step, the unions re-composed on the "
                                               "device inside it; fields gathered on the device per step (inside the timed region)"}})
        # ---- the reference's variable-coordinate loop (static_trainer.py:180-202 inside optimizers.py:247-257), UNCHANGED -- per-step upload of the
        # fields from host memory, zero_grad, eager call with coordinates and per-sample graphs, nn.MSELoss, backward, torch.optim.AdamW -- under
        # autograph: (a) one fixed batch, (b) a shuffling loader over the resident dataset, (c) the same with the graphs uploaded anew every step
        # from host memory (what move_to_device does there: new dict objects, new tensors every step)
        from gaot_amd import ops as _o
        _o.register_grad_slots([], [])
        torch.manual_seed(0)
        m2 = mk()
        opt = torch.optim.AdamW(m2.parameters(), lr=8e-4, weight_decay=1e-5)
        lossf = torch.nn.MSELoss()
        enc_host, dec_host = csr(enc_all), csr(dec_all)
        up = lambda rows, b: [[{"neighbors_index": c[0].to(dev), "neighbors_row_splits": c[1].to(dev)} for c in rows[i]] for i in b]

        def host_batch(b):
            """what the loader's workers hand the training loop (collate_variable_batch, data_utils.py:272-294): host tensors of one batch.
            Assembled BEFORE the timed region, as a prefetching DataLoader does -- on this box a multi-threaded CPU gather of 1.5 MB costs
            20-100 ms (the container's CPU quota against torch's intra-op threads), which is the loader's time, not the step's."""
            return (p_all[b], t_all[b], x_all[b], b)

        def loop_step(hb, upload=False):
            xb, yb, xc = hb[0].to(dev), hb[1].to(dev), hb[2].to(dev)
            b = hb[3]
            e_, d_ = (up(enc_host, b), up(dec_host, b)) if upload else (pick(enc_all, b), pick(dec_all, b))
            opt.zero_grad()
            out_ = m2(latent_tokens_coord=latd, xcoord=xc, pndata=xb, encoder_nbrs=e_, decoder_nbrs=d_)
            lossf(out_, yb).backward()
            opt.step()
            return type(out_.grad_fn).__name__ == "_GraphedStepBackward"

        def timed(batches, n_warm, **kw):
            for hb in batches[:n_warm]:
                loop_step(hb, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            graphed = sum(loop_step(hb, **kw) for hb in batches[n_warm:])
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / (len(batches) - n_warm), int(graphed)
        b0 = list(range(B))
        for key, batches, n_warm, kw, what in (
                ("reference_loop_vx", [host_batch(b0)] * (6 + steps), 6, {},
                 "the reference's vx loop unchanged (host->device upload of fields and coordinates per step, zero_grad, eager call, nn.MSELoss, "
                 "torch.optim.AdamW), ONE batch of resident per-sample graphs every step: forward and backward replay as hipGraphs (autograph.py)"),
                ("reference_loop_vx_shuffled", [host_batch(draw()) for _ in range(3 * warmup + 9 + steps)], 3 * warmup + 9, {},
                 "the same loop under a SHUFFLING loader (the reference's default): a new composition of 16 of the 48 resident meshes every step; "
                 "the captured forward re-composes the unions on the device from a per-step table (plan.StaticUnion)"),
                ("reference_loop_vx_uploaded", [host_batch(draw()) for _ in range(6 + steps)], 6, {"upload": True},
                 "... with the per-sample graphs uploaded anew from host memory every step (move_to_device, static_trainer.py:192-193: 32 new "
                 "dicts per step): nothing is built per sample -- the captured forward composes the unions from the raw int64 lists and derives the "
                 "decoder's transposed CSR on the device; what remains above the shuffled leg is the 64 uploads themselves")):
            dt_, graphed = timed(batches, n_warm, **kw)
            out["C3"][key] = {"value": B / dt_, "unit": "samples/s", "ms_per_step": 1e3 * dt_, "steps": steps, "graphed_steps": graphed,
                              "frac_of_trainstep": (B / dt_) / out["C3"]["samples_per_s"], "what": what}
        del m2, opt
    if "C4" in which:
        torch.manual_seed(0)
        B, N = 4, N_NODES
        mc = MAGNOConfig(radius=RADIUS, lifting_channels=C_LIFT)
        model = GAOT(4, 2, NS(args=NS(magno=mc, transformer=TransformerConfig(patch_size=PATCH, hidden_size=HIDDEN)), latent_tokens_size=LATENT)).to(dev).train()
        g = torch.Generator().manual_seed(4)
        lat = grid(LATENT)
        x = torch.rand(N, 2, generator=torch.Generator().manual_seed(0)) * 2 - 1
        xb, tgt = torch.randn(B, N, 4, generator=g), torch.randn(B, N, 2, generator=g)
        from oracle import gaot_oracle as O
        ocfg = O.OracleConfig(radius=RADIUS, hidden_size=64, lifting_channels=C_LIFT, patch_size=PATCH, tf_hidden_size=HIDDEN,
                              latent_tokens_size=LATENT, precompute_edges=True)
        enc, dec = [O.radius_csr(x, lat, RADIUS, exact=True)], [O.radius_csr(lat, x, RADIUS, exact=True)]
        Q = LATENT[0] * LATENT[1]
        measure("C4", model, xb.to(dev), tgt.to(dev), dict(latent_tokens_coord=lat.to(dev), xcoord=x.to(dev)), ocfg,
                dict(latent=lat, xcoord=x, pndata=xb, encoder_nbrs=enc, decoder_nbrs=dec), B,
                f"BASELINE configs[3]: NS-Gauss-shaped time-dependent 2D, {N} nodes, batch {B} per GPU, in = u(2) + 2 time columns, out 2: "
                "pair-training step + 10-step autoregressive rollout (stepper 'time_der')",
                (int(enc[0][0].numel()), int(dec[0][0].numel()), Q, N, C_LIFT, B, 4, 2))
        model.eval()
        stats = {"u": {"mean": torch.zeros(2), "std": torch.ones(2)}, "der": {"mean": torch.zeros(2), "std": torch.ones(2)},
                 "start_time": {"mean": 0.0, "std": 1.0}, "time_diffs": {"mean": 0.0, "std": 1.0}}
        tv, ti = np.linspace(0, 1, 21), np.arange(0, 22, 2)[:11]
        xbd, latd, xd = xb.to(dev), lat.to(dev), x.to(dev)
        roll = lambda: model.autoregressive_predict(x_batch=xbd[..., :2], time_indices=ti, t_values=tv, stats=stats, stepper_mode="time_der",
                                                    latent_tokens_coord=latd, fixed_coord=xd)
        for _ in range(2):
            roll()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            roll()
        torch.cuda.synchronize()
        out["C4"]["rollout_ms"] = 1e3 * (time.perf_counter() - t0) / 5
        out["C4"]["rollout_steps"] = 10
    if "C5" in which:
        torch.manual_seed(0)
        B, N = 1, 65536
        lat3 = [32, 32, 32]
        mc = MAGNOConfig(coord_dim=3, radius=0.067, lifting_channels=48)
        model = GAOT(3, 1, NS(args=NS(magno=mc, transformer=TransformerConfig(patch_size=2, hidden_size=384, attn_config=AttentionConfig(num_heads=8, num_kv_heads=8))),
                              latent_tokens_size=lat3)).to(dev).train()
        g = torch.Generator().manual_seed(5)
        lat = grid(lat3)
        x = shell_points(N, torch.Generator().manual_seed(0))
        p, t = torch.randn(B, N, 3, generator=g), torch.randn(B, N, 1, generator=g)
        from oracle import gaot_oracle as O
        ocfg = O.OracleConfig(coord_dim=3, radius=0.067, hidden_size=64, lifting_channels=48, patch_size=2, tf_hidden_size=384,
                              num_heads=8, num_kv_heads=8, latent_tokens_size=lat3, precompute_edges=True)
        kw = dict(latent_tokens_coord=lat.to(dev), xcoord=x.to(dev))
        with torch.no_grad():
            model(pndata=p.to(dev), **kw)            # builds the radius graphs (HIP cell list); the oracle integrates over the same lists
        nbe = list(model.encoder.neighbor_cache.values())[0][0]
        nbd = list(model.decoder.neighbor_cache.values())[0][0]
        enc = [(nbe["neighbors_index"].cpu(), nbe["neighbors_row_splits"].cpu())]
        dec = [(nbd["neighbors_index"].cpu(), nbd["neighbors_row_splits"].cpu())]
        deg = enc[0][1][1:] - enc[0][1][:-1]
        measure("C5", model, p.to(dev), t.to(dev), kw, ocfg, dict(latent=lat, xcoord=x, pndata=p, encoder_nbrs=enc, decoder_nbrs=dec), B,
                f"BASELINE configs[4]: synthetic 3D surface cloud (three ellipsoid shells), {N} nodes, batch {B} per GPU, latent 32^3 = 4096 tokens of "
                f"384 (head_dim 48), 48 lifting channels; {int(enc[0][0].numel())} encoder edges, degree max {int(deg.max())}, {int((deg == 0).sum())} empty latent rows",
                (int(enc[0][0].numel()), int(dec[0][0].numel()), 32 ** 3, N, 48, B, 3, 1))
    return out


def hip_reference_pass(ts, model, tensors):
    """prediction, loss and every gradient of the HIP path at the INITIAL weights, through the TRAINING path itself: TrainStep's eager
    forward + backward (the same launches the captured step replays, including the grouped weight-gradient launch), before any update"""
    lat, x, p, t = tensors
    with torch.no_grad():
        pred = model(latent_tokens_coord=lat, xcoord=x, pndata=p)
    loss = ts._forward_backward()
    torch.cuda.synchronize()
    csr = lambda m: tuple(list(m.neighbor_cache.values())[0][0][k].cpu() for k in ("neighbors_index", "neighbors_row_splits"))
    out = {"pred": pred.detach().cpu(), "loss": float(loss), "grads": {k: q.grad.detach().cpu().clone() for k, q in model.named_parameters()},
           "enc_csr": csr(model.encoder), "dec_csr": csr(model.decoder)}
    ts.bucket.clear()
    return out


def self_launch(n: int) -> None:
    """re-run this command line under `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` on a free loopback port and pass
    its stdout / exit status through (stderr of the ranks goes to ours)"""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL / tensor sharing across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    r = subprocess.run(cmd, env=env)
    if r.returncode != 0:
        sys.exit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-loop", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary C3 / C4 / C5 measurements")
    ap.add_argument("--dtype", choices=["f32", "bf16x2", "bf16"], default="f32",
                    help="f32 (default, the parity path and the only headline): exact three-piece products.  bf16x2 / bf16: run the whole "
                         "line at that narrower arithmetic (A/B and profiling runs; the line says so in `dtype`).  The default line "
                         "already carries both as `variants`.")
    ap.add_argument("--no-variants", action="store_true", help="skip the bf16x2 / bf16 variant measurements")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N` (no launcher): start the N ranks ourselves, one process per GPU, exactly as the documented
        # launch line does; rank 0's single JSON line is the only thing the children print to stdout
        return self_launch(args.gpus)
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} disagrees with WORLD_SIZE={world} (launch with --nproc-per-node {args.gpus}, or run without a launcher)")
    # test hooks (1-GPU boxes only): run the N > 1 code path with every rank on device 0 over gloo
    if os.environ.get("GAOT_BENCH_FORCE_DEVICE") is not None:
        local = int(os.environ["GAOT_BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("GAOT_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)      # RCCL over xGMI
        else:
            dist.init_process_group(backend=backend)

    from gaot_amd.trainer import TrainStep
    from gaot_amd import ops as _ops
    if args.dtype == "bf16":
        from gaot_amd import _lib
        _lib.load().gaot_debug_set_gemm_pieces(1)      # the tile kernels the heuristic picks, with ONE bf16 piece per operand
        os.environ["GAOT_BENCH_ONE_PIECE"] = "1"
    elif args.dtype == "bf16x2":
        _ops.set_precision("bf16x2")
    torch.manual_seed(0)                      # identical weights on every rank (and broadcast from rank 0 anyway)
    model = build_model().to(dev).train()
    lat, x, p, t = synthetic(1234 + rank, dev)
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    sd0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()} if want_cpu else None
    staged = {"0": False, "1": True}.get(os.environ.get("GAOT_BENCH_STAGED", ""), None)     # A/B hook; default: staged when world > 1
    ts = TrainStep(model, lr=8e-4, weight_decay=1e-5, use_graph=not args.no_graph, staged=staged)
    ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
    hip0 = hip_reference_pass(ts, model, (lat, x, p, t)) if want_cpu else None

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        ts.step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts.step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    loss = float(ts._loss if ts.use_graph else ts.step())
    comm = None
    ranks_seen = None
    if world > 1:
        # proof that the collective saw N distinct ranks / devices and that they stayed identical: every rank contributes its device index
        # and a checksum of its flat parameter buffer after the timed steps (one all-gather, after the timed region)
        flat_w = ts.opt.flat_p if hasattr(ts.opt, "flat_p") else torch.cat([q.detach().reshape(-1) for q in model.parameters()])
        mine = torch.tensor([float(rank), float(torch.cuda.current_device()), float(flat_w.double().sum()), float(flat_w.double().abs().sum())],
                            device=dev, dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows = [[float(v) for v in r_.tolist()] for r_ in allr]
        ranks_seen = {"n_ranks_seen": len({int(r_[0]) for r_ in rows}), "devices": [int(r_[1]) for r_ in rows],
                      "param_checksums_identical": all(r_[2:] == rows[0][2:] for r_ in rows), "param_checksum": rows[0][2:]}
        # exposed communication = the same K steps WITHOUT the gradient exchange (ranks drift apart: timing only, after the headline)
        # subtracted from the headline; plus every phase slice's all-reduce timed on its own (RCCL over xGMI, HIP events)
        slices = []
        for k, ks in enumerate(ts.stage_groups if ts.staged else [list(range(ts.bucket.n_phases))]):
            lo, hi = ts.bucket.segments[ks[0]][0], ts.bucket.segments[ks[-1]][1]
            buf = ts.bucket.flat[lo:hi]
            for _ in range(3):
                dist.all_reduce(buf, op=dist.ReduceOp.AVG if dist.get_backend() == "nccl" else dist.ReduceOp.SUM)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10):
                dist.all_reduce(buf, op=dist.ReduceOp.AVG if dist.get_backend() == "nccl" else dist.ReduceOp.SUM)
            b.record()
            torch.cuda.synchronize()
            slices.append({"stage_group": k, "backward_phases": list(ks), "bytes": 4 * (hi - lo), "allreduce_us_alone": 1e3 * a.elapsed_time(b) / 10})
        ts.comm_enabled = False
        for _ in range(3):
            ts.step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ts.step()
        sync()
        no_comm = time.perf_counter() - t0
        ts.comm_enabled = True
        tt = torch.tensor([no_comm], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        no_comm = float(tt.item())
        comm = {"backend": dist.get_backend() + (" (RCCL over xGMI)" if dist.get_backend() == "nccl" else ""), **ranks_seen, "slices": slices,
                "ms_per_step_without_exchange": 1e3 * no_comm / args.steps,
                "exposed_ms_per_step": 1e3 * (elapsed - no_comm) / args.steps,
                "what": "per stage group (a run of backward phases sharing one graph and one grouped weight-gradient launch) one asynchronous all-reduce "
                        "(ReduceOp.AVG) of its slice of the flat gradient buffer, issued while the next group's backward graph replays; exposed = "
                        "headline step - the same step without the exchange"}

    roof = gemm_roofline(ts)
    if rank == 0:
        n_params = sum(q.numel() for q in model.parameters())
        ms_step = 1e3 * elapsed / args.steps
        traffic, traffic_src = recorded_traffic("gemm")
        # HBM regime: the four fused integral-transform kernels of one step.  Algorithmic bytes = every operand
        # of a launch touched once (k_e rows, node features, CSR arrays, outputs) -- fewer than SURVEY 8d's 42 MB per train
        # sample because the lifted [B,N,C] tensor and the [B,Nq,C] transform output no longer exist.
        enc_nb = list(model.encoder.neighbor_cache.values())[0][0]
        dec_nb = list(model.decoder.neighbor_cache.values())[0][0]
        Ee, Ed = int(enc_nb["neighbors_index"].numel()), int(dec_nb["neighbors_index"].numel())
        Q, C, B_ = LATENT[0] * LATENT[1], C_LIFT, BATCH
        gno_bytes = {
            "encoder fwd": 4.0 * (Ee * C + B_ * N_NODES * 1 + 3 * Ee + Q + B_ * Q * C),
            "encoder bwd": 4.0 * (B_ * Q * C + Ee * C + B_ * N_NODES * 1 + 3 * Ee + Ee * C),
            "decoder fwd": 4.0 * (Ed * C + B_ * Q * C + 3 * Ed + N_NODES + B_ * N_NODES * 1),
            "decoder bwd": 4.0 * (B_ * N_NODES * 1 + Ed * C + B_ * Q * C + 5 * Ed + Ed * C + B_ * Q * C),
        }
        gno_rows = {k: {"us": roof["gno_us"].get(k), "algorithmic_MB": gno_bytes[k] / 1e6,
                        "GBps": (gno_bytes[k] / (roof["gno_us"][k] * 1e-6) / 1e9) if roof["gno_us"].get(k) else None} for k in gno_bytes}
        gno_total_us = sum(v for k, v in roof["gno_us"].items() if k in gno_bytes)
        gno_total_b = sum(gno_bytes.values())
        hbm_achieved = gno_total_b / (gno_total_us * 1e-6) / 1e9 if gno_total_us > 0 else 0.0
        gno_traffic, gno_src = recorded_traffic("gno")
        n_prod = {"f32": 3.0 if _ops._F16_PIECES[0] else 6.0, "bf16x2": 3.0, "bf16": 1.0}[args.dtype]
        t_mfma_ideal = n_prod * 259.0e9 / (PEAK_BF16_MATRIX_TFLOPS * 1e12) * 1e3   # 259 GFLOP per B = 8 step, as piece products on the pipe in use
        t_hbm_ideal = (8 * (42 + 120) + 13) * 1e6 / (PEAK_HBM_GBPS * 1e9) * 1e3   # ~1.31 GB per step
        line = {
            "metric": "train samples/sec (2D 16k-node mesh, bs=8 per GPU)",
            "value": BATCH * world * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x2": "bf16x2 products (two rounded bf16 pieces per f32 operand), f32 accumulate -- NOT the parity headline",
                      "bf16": "bf16 tile-GEMM operands (one piece), f32 accumulate -- NOT the parity headline"}[args.dtype],
            "precision": {"storage_and_accumulation": "f32 everywhere (weights, activations, gradients, optimizer state, every accumulator)",
                          "pieces": dict(_ops._PIECES), "mode": _ops.precision() if args.dtype != "bf16" else "bf16 (one piece in the tile GEMMs)",
                          "f32_pieces": "fp16x2" if _ops._F16_PIECES[0] else "bf16x3",
                          "what": "`pieces` 3 = the f32 precision (the headline): every operand of a matrix-pipe product is carried at fp32 width -- "
                                  "f32_pieces fp16x2: as two fp16 pieces h + m of the operand scaled by a power of two from its device-resident magnitude "
                                  "word (|s x - h - m| <= 2^-23 |s x|: at most the operand's last bit, zero for three values in four; three piece products h h + h m + m h) in the GEMM tiles, the "
                                  "grouped weight gradients and the attention for head_dim <= 64; as three bf16 pieces (8 + 8 + 8 bits exactly, six piece "
                                  "products) in the kernel MLP, wherever a product has no magnitude word, and everywhere with GAOT_F32_PIECES=bf16x3 "
                                  "(`variants.bf16x3`).  Measured against float64 the fp16x2 products are at or below the bf16x3 ones (tests/test_ops_gpu.py "
                                  "test_gemm_fp16_piece_products, test_attention_fp16_piece_products; `rel_l2_vs_oracle.vs_float64_oracle` here).  `pieces` 2 = two "
                                  "rounded bf16 pieces, 16 significant bits: `variants.bf16x2`, never `value`"},
            "data": "synthetic (uniform-random 16384-point 2-D mesh in [-1,1]^2, N(0,1) fields, random-init weights)",
            "config": {"workload": "BASELINE configs[1]: Poisson-Gauss-shaped 2D, 16384 nodes/mesh, batch 8 per GPU, fx mode; "
                                   "example model latent 64x64, C=64, patch 2, transformer 256x3, 8 heads",
                       "params": n_params, "global_batch": BATCH * world, "parallelism": f"dp{world}",
                       "step": "fwd + MSE + bwd + AdamW" + (" + staged flat-grad RCCL all-reduce (one async slice per stage group)" if world > 1 else ""),
                       "h2d": "excluded: the timed region replays one batch bound in HBM ahead of time (TrainStep.bind); nothing is uploaded or "
                              "re-copied per step (the PCIe-inclusive drop-in rate is `reference_loop`)",
                       "hipgraph": ts.use_graph, "staged_backward_phases": ts.bucket.n_phases, "stage_groups": ts.stage_groups, "final_loss": loss},
            "roofline": {"bound": "mfma", "pipe": "f16 / bf16 mfma (v_mfma_f32_32x32x16_f16 / _bf16: the same 2.5 PFLOP/s dense rate)",
                         "kernel": "every launch of one step on the bf16 matrix pipe behind gaot_gemm_f32: gemm_split_kernel / gemm_ad_kernel (NT / NN tiles: operands staged through registers / both by LDS DMA) and the grouped "
                                   "weight-gradient launch gemm_tn_grouped_kernel; each f32 operand as "
                                   f"{ {'f32': 'two fp16 pieces of the power-of-two scaled operand (24 significant bits up to its last one), three piece products on v_mfma_f32_32x32x16_f16' if _ops._F16_PIECES[0] else 'three exact bf16 pieces, six piece products', 'bf16x2': 'two rounded bf16 pieces, three piece products', 'bf16': 'one bf16 piece, one product'}[args.dtype] }",
                         "achieved": roof["tflops"], "peak": PEAK_BF16_MATRIX_TFLOPS, "unit": "TFLOP/s",
                         "frac": roof["tflops"] / PEAK_BF16_MATRIX_TFLOPS,
                         "achieved_note": "ALGORITHMIC FLOPs (2MNK per product, each counted once whatever the number of piece products) / event-timed "
                                          "duration of the family's launches, against the dense f16 / bf16 matrix peak the kernels run on; the pipe "
                                          "itself is issued three piece products per product: `issued_piece_tflops`",
                         "issued_piece_tflops": roof["piece_tflops"], "issued_frac": roof["piece_tflops"] / PEAK_BF16_MATRIX_TFLOPS,
                         "f32_equivalent_frac_of_f32_matrix_peak": roof["tflops"] / PEAK_F32_MATRIX_TFLOPS,
                         "traffic": traffic,
                         "traffic_note": "HBM bytes per launch, PMC FETCH_SIZE(x2, gfx950)+WRITE_SIZE, separate --pmc passes over the same "
                                         f"launches (eager step): {traffic_src}",
                         "algorithmic_bytes_per_launch": roof["alg_bytes_per_launch"],
                         "launches_per_step": roof["launches"], "gflop_per_step": roof["flops"] / 1e9, "piece_gflop_per_step": roof["piece_flops"] / 1e9,
                         "kernel_ms_per_step": roof["ms"], "avg_launch_us": 1e3 * roof["ms"] / max(1, roof["launches"]),
                         "f32_mfma_tiles": {"launches_per_step": roof["f32_mfma_launches"], "ms_per_step": roof["f32_mfma_ms"], "gflop_per_step": roof["f32_mfma_gflop"],
                                            "note": "products too small / narrow for the split tiles run on v_mfma_f32_32x32x2_f32 (157.3 TFLOP/s dense): not part of `achieved`"},
                         "skinny_valu_launches_per_step": roof["skinny_launches"], "all_gemm_entry_ms_per_step": roof["all_gemm_ms"]},
            "roofline_hbm": {"bound": "hbm", "kernel": "fused integral-transform kernels of one step (csrc/gno.hip, gno_ep.hip): lift_gather_reduce, "
                                                        "lift_edge_grad, proj_fwd_bin, proj_edge_grad + proj_gather_t (row-parallel forms: the bench "
                                                        "mesh is not degree-skewed; skewed plans take the edge-partitioned forms)",
                             "achieved": hbm_achieved, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": hbm_achieved / PEAK_HBM_GBPS,
                             "traffic": gno_traffic, "traffic_note": f"HBM bytes of the family per step, PMC FETCH_SIZE(x2)+WRITE_SIZE: {gno_src}",
                             "algorithmic_bytes_per_step": gno_total_b, "kernel_us_per_step": gno_total_us, "per_kernel": gno_rows,
                             "edges": {"encoder": Ee, "decoder": Ed}},
            "roofline_step": {"t_mfma_ideal_ms": t_mfma_ideal, "t_hbm_ideal_ms": t_hbm_ideal,
                              "achieved": max(t_mfma_ideal, t_hbm_ideal) / ms_step,
                              "note": "SURVEY 8d: max(t_HBM, t_MFMA)_ideal / t_measured with 259 GFLOP (x piece products per product, on the bf16 "
                                      "matrix pipe at 2500 TFLOP/s) and ~1.31 GB of algorithmic work per 8-sample step at 8 TB/s"},
        }
        if comm is not None:
            line["comm"] = comm
        if world == 1:
            # sustained figure: the same step for at least one second of wall time (the headline window is K steps ~ 0.15 s)
            n_sus = max(args.steps, int(1.2 / max(ms_step * 1e-3, 1e-6)))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_sus):
                ts.step()
            torch.cuda.synchronize()
            dt_sus = time.perf_counter() - t0
            line["sustained"] = {"value": BATCH * n_sus / dt_sus, "unit": "samples/s", "steps": n_sus, "seconds": dt_sus,
                                 "ms_per_step": 1e3 * dt_sus / n_sus}
        if world == 1 and not args.no_configs:
            del ts
            line["configs"] = secondary_configs(dev, oracle=not args.no_cpu_baseline)
        if world == 1 and not args.no_reference_loop:
            line["reference_loop"] = reference_loop_rate(dev)
            line["reference_loop"]["frac_of_headline"] = line["reference_loop"]["value"] / line["value"]
        ref64 = None
        if want_cpu:
            line["cpu_baseline"], line["rel_l2_vs_oracle"], ref64 = cpu_baseline(sd0, (lat, x, p, t), hip0)
            try:
                line["torch_gpu_baseline"] = torch_gpu_baseline(sd0, (lat, x, p, t), hip0, dev)
                line["torch_gpu_baseline"]["headline_over_this"] = line["value"] / line["torch_gpu_baseline"]["value"]
            except Exception as e:          # a baseline leg must never cost the line
                line["torch_gpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()
        if world == 1 and not args.no_variants and args.dtype == "f32":
            # narrower arithmetic than the reference's fp32: labelled variants next to the headline, each with its own error vs float64
            line["variants"] = {nm: variant_rate(dev, nm, ref64) for nm in ("bf16x3", "bf16x2", "bf16")}
            for v in line["variants"].values():
                v["speedup_vs_headline"] = v["value"] / line["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
