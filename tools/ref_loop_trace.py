#!/usr/bin/env python3
"""The reference-shaped loop of bench.py (resident tensors or per-step uploads) for a kernel trace: ref_loop_trace.py [upload] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
upload = "upload" in sys.argv
torch.manual_seed(0)
model = bench.build_model().to(dev).train()
lat, x, p, t = bench.synthetic(1234, torch.device("cpu"))
if not upload: lat, x, p, t = [v.to(dev) for v in (lat, x, p, t)]
opt = torch.optim.AdamW(model.parameters(), lr=8e-4, weight_decay=1e-5)
loss_fn = torch.nn.MSELoss()
def one():
    xb, yb, latd, coord = (p.to(dev), t.to(dev), lat.to(dev), x.to(dev)) if upload else (p, t, lat, x)
    opt.zero_grad()
    loss = loss_fn(model(latent_tokens_coord=latd, xcoord=coord, pndata=xb), yb)
    loss.backward()
    opt.step()
for _ in range(8): one()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n): one()
torch.cuda.synchronize(); print(f"{'upload' if upload else 'resident'}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per step")
