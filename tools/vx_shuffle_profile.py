#!/usr/bin/env python3
"""Where the time of the reference's vx loop goes under a SHUFFLING loader (bench.py configs.C3.reference_loop_vx_shuffled): cProfile of the
host side over a few steps + wall time per step."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
import torch
from gaot_amd.model.gaot import GAOT
from gaot_amd.model.layers.magno import MAGNOConfig
from gaot_amd.model.layers.attn import TransformerConfig
from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch
from tests._workloads import grid, naca_points
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, N = 16, 8192
mc = MAGNOConfig(radius=0.033, lifting_channels=64, precompute_edges=True)
m2 = GAOT(3, 1, NS(args=NS(magno=mc, transformer=TransformerConfig(patch_size=2, hidden_size=256)), latent_tokens_size=[64, 64])).to(dev).train()
g = torch.Generator().manual_seed(0)
latd = grid([64, 64]).to(dev)
x = torch.stack([naca_points(N, g, 0.15) for _ in range(B)])
p, t = torch.randn(B, N, 3, generator=g), torch.randn(B, N, 1, generator=g)
ns = NeighborSearch("auto")
xd = x.to(dev)
enc = [[ns(xd[b], latd, 0.033)] for b in range(B)]
dec = [[ns(latd, xd[b], 0.033)] for b in range(B)]
opt = torch.optim.AdamW(m2.parameters(), lr=8e-4, weight_decay=1e-5)
lossf = torch.nn.MSELoss()
gsh = torch.Generator().manual_seed(7)
def one():
    perm = torch.randperm(B, generator=gsh).tolist()
    xb, yb = p[perm].to(dev), t[perm].to(dev)
    opt.zero_grad()
    out_ = m2(latent_tokens_coord=latd, xcoord=xd[perm], pndata=xb, encoder_nbrs=[enc[i] for i in perm], decoder_nbrs=[dec[i] for i in perm])
    lossf(out_, yb).backward()
    opt.step()
import gc
if "--nogc" in sys.argv: gc.disable()
for _ in range(4): one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): one()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 5 * 1e3:.1f} ms per shuffled step", flush=True)
if "--leak" in sys.argv:
    import gc, collections
    def snapshot():
        gc.collect()
        c = collections.Counter()
        for o in gc.get_objects():
            try:
                if isinstance(o, torch.Tensor) and o.is_cuda:
                    c[(tuple(o.shape), str(o.dtype))] += 1
            except Exception:
                pass
        return c
    for _ in range(3): one()
    torch.cuda.synchronize(); a = snapshot(); m0 = torch.cuda.memory_allocated()
    for _ in range(4): one()
    torch.cuda.synchronize(); b = snapshot(); m1 = torch.cuda.memory_allocated()
    print("allocated grew by", (m1 - m0) / 2**20, "MiB over 4 steps", flush=True)
    grown = [(k, b[k] - a.get(k, 0)) for k in b if b[k] - a.get(k, 0) > 0]
    grown.sort(key=lambda kv: -kv[1] * (torch.tensor(kv[0][0]).prod().item() if kv[0][0] else 1))
    for k, n in grown[:25]:
        print(n, "more of", k, flush=True)
    big = [o for o in gc.get_objects() if isinstance(o, torch.Tensor) and o.is_cuda and tuple(o.shape) == grown[0][0][0]]
    for o in big[:2]:
        for r in gc.get_referrers(o)[:6]:
            print("  referrer:", type(r).__name__, (list(r.keys())[:8] if isinstance(r, dict) else (len(r) if hasattr(r, "__len__") else "")), flush=True)
    sys.exit(0)
if "--slow" in sys.argv:
    import io
    for k in range(9):
        pr = cProfile.Profile(); pr.enable()
        t0 = time.perf_counter(); one(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        pr.disable()
        print(f"step {k}: {dt * 1e3:.1f} ms", flush=True)
        if dt > 0.03:
            buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(6)
            print("\n".join(l for l in buf.getvalue().splitlines() if "/" in l or "{" in l), flush=True)
    sys.exit(0)
if "--alloc" in sys.argv:
    for k in range(8):
        st0 = torch.cuda.memory_stats()
        t0 = time.perf_counter(); one(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st1 = torch.cuda.memory_stats()
        print(f"step {k}: {dt * 1e3:.1f} ms  device allocs +{st1['num_device_alloc'] - st0['num_device_alloc']} frees +{st1['num_device_free'] - st0['num_device_free']} "
              f"reserved {st1['reserved_bytes.all.current'] / 2**20:.0f} MiB  allocated peak {st1['allocated_bytes.all.peak'] / 2**20:.0f} MiB  retries {st1['num_alloc_retries']}", flush=True)
    sys.exit(0)
if "--cycles" in sys.argv:
    import gc, collections
    gc.collect()
    gc.set_debug(gc.DEBUG_SAVEALL)
    one(); torch.cuda.synchronize()
    n = gc.collect()
    print("unreachable objects after ONE step:", n, flush=True)
    cnt = collections.Counter(type(o).__name__ for o in gc.garbage)
    print(cnt.most_common(25), flush=True)
    tens = [o for o in gc.garbage if isinstance(o, torch.Tensor)]
    print("tensors in cycles:", len(tens), "bytes", sum(t.numel() * t.element_size() for t in tens if t.is_cuda), flush=True)
    for o in gc.garbage:
        if type(o).__name__ in ("GeometryPlan", "MergedGeometry"):
            refs = [type(r).__name__ + (":" + str(list(r.keys())[:6]) if isinstance(r, dict) else "") for r in gc.get_referrers(o) if r is not gc.garbage]
            print(type(o).__name__, "referred by", refs[:8], flush=True)
    fns = [o for o in gc.garbage if "Backward" in type(o).__name__ or type(o).__name__ == "function" or type(o).__name__ == "cell"]
    for o in fns[:12]:
        print(type(o).__name__, getattr(o, "__qualname__", ""), flush=True)
    sys.exit(0)
import gc
gc_t = [0.0, 0]
def _cb(phase, info):
    if phase == "start": _cb.t0 = time.perf_counter()
    else: gc_t[0] += time.perf_counter() - _cb.t0; gc_t[1] += 1
gc.callbacks.append(_cb)
def phases():
    T = {}
    def mark(name, t0):
        T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    for _ in range(5):
        t0 = time.perf_counter(); perm = torch.randperm(B, generator=gsh).tolist(); xb, yb = p[perm].to(dev), t[perm].to(dev); xs = xd[perm]; mark("upload+index", t0)
        t0 = time.perf_counter(); opt.zero_grad(); mark("zero_grad", t0)
        t0 = time.perf_counter(); out_ = m2(latent_tokens_coord=latd, xcoord=xs, pndata=xb, encoder_nbrs=[enc[i] for i in perm], decoder_nbrs=[dec[i] for i in perm]); mark("forward host", t0)
        t0 = time.perf_counter(); torch.cuda.synchronize(); mark("forward gpu wait", t0)
        t0 = time.perf_counter(); l = lossf(out_, yb); l.backward(); mark("backward host", t0)
        t0 = time.perf_counter(); torch.cuda.synchronize(); mark("backward gpu wait", t0)
        t0 = time.perf_counter(); opt.step(); mark("opt host", t0)
        t0 = time.perf_counter(); torch.cuda.synchronize(); mark("opt gpu wait", t0)
    print({k: round(v / 5 * 1e3, 2) for k, v in T.items()}, "ms per step; gc:", round(gc_t[0] / 5 * 1e3, 2), "ms per step in", gc_t[1], "collections", flush=True)
phases()
gc_t[0] = 0.0; gc_t[1] = 0
phases()
sys.exit(0)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): one()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(25)
