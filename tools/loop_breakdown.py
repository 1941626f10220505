#!/usr/bin/env python3
"""Where the reference-shaped training loop (bench.py reference_loop) loses time against the TrainStep headline:
variants of the same C2 step, each timed over 30 steps after 8 warm-up steps.  usage: python tools/loop_breakdown.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaot_amd import ops
from gaot_amd.trainer import TrainStep

dev = torch.device("cuda:0")


def timed(fn, warm=8, iters=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / iters


def fresh():
    ops.register_grad_slots([], [])
    torch.manual_seed(0)
    return bench.build_model().to(dev).train()


lat_c, x_c, p_c, t_c = bench.synthetic(1234, torch.device("cpu"))
lat, x, p, t = [v.to(dev) for v in (lat_c, x_c, p_c, t_c)]
rows = []

for graph in (True, False):
    m = fresh()
    ts = TrainStep(m, use_graph=graph)
    ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
    rows.append((f"TrainStep hipGraph={graph}", timed(ts.step)))


def loop(upload_batch, upload_coords, optimizer, loss_kind, host_only=False):
    m = fresh()
    opt = torch.optim.AdamW(m.parameters(), lr=8e-4, weight_decay=1e-5, **optimizer)
    lossf = torch.nn.MSELoss() if loss_kind == "torch" else ops.mse_loss

    def one():
        xb, yb = (p_c.to(dev), t_c.to(dev)) if upload_batch else (p, t)
        latd, coord = (lat_c.to(dev), x_c.to(dev)) if upload_coords else (lat, x)
        opt.zero_grad()
        loss = lossf(m(latent_tokens_coord=latd, xcoord=coord, pndata=xb), yb)
        loss.backward()
        opt.step()
    return one


rows.append(("eager loop, resident tensors, torch AdamW (foreach), torch MSELoss", timed(loop(False, False, {}, "torch"))))
rows.append(("  + HIP mse_loss", timed(loop(False, False, {}, "hip"))))
rows.append(("  + fused=True AdamW", timed(loop(False, False, {"fused": True}, "torch"))))
rows.append(("eager loop, batch uploaded per step", timed(loop(True, False, {}, "torch"))))
p_pag, t_pag = p_c, t_c
p_c, t_c = p_c.pin_memory(), t_c.pin_memory()          # what the reference's DataLoader hands over (data_processor.py:357: pin_memory=True)
rows.append(("eager loop, batch uploaded per step from PINNED memory", timed(loop(True, False, {}, "torch"))))
rows.append(("eager loop, pinned batch AND pageable coordinates uploaded per step", timed(loop(True, True, {}, "torch"))))
p_c, t_c = p_pag, t_pag
rows.append(("eager loop, batch AND coordinates uploaded per step (= bench reference_loop)", timed(loop(True, True, {}, "torch"))))

# host-side cost of one eager step: enqueue only (GPU parked behind a long sleep)
m = fresh()
opt = torch.optim.AdamW(m.parameters(), lr=8e-4, weight_decay=1e-5)
one = loop(False, False, {}, "torch")
for _ in range(5):
    one()
torch.cuda.synchronize()
torch.cuda._sleep(int(0.2 * 2.0e9))
t0 = time.perf_counter()
for _ in range(10):
    one()
host = 1e3 * (time.perf_counter() - t0) / 10
torch.cuda.synchronize()
rows.append(("host enqueue time of one eager step (GPU parked)", host))
for name, ms in rows:
    print(f"{ms:8.3f} ms  {name}")
