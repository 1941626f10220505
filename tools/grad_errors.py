#!/usr/bin/env python3
"""Per-tensor gradient error of the HIP path at the bench configuration against the oracle evaluated in FLOAT64 (weights, batch
and every intermediate in double): separates the kernels' own rounding from the fp32 oracle's.  Prints the worst tensors."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from oracle import gaot_oracle as O
dev = torch.device("cuda:0")
if len(sys.argv) > 3:      # grad_errors.py <rows> <debug hook> <value>
    from gaot_amd import _lib
    print(sys.argv[2], sys.argv[3], "previous:", getattr(_lib.load(), sys.argv[2])(int(sys.argv[3])))
torch.manual_seed(0)
model = bench.build_model().to(dev).train()
tensors = bench.synthetic(1234, dev)
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
from gaot_amd.trainer import TrainStep
ts = TrainStep(model, lr=8e-4, weight_decay=1e-5, use_graph=False)
ts.bind(tensors[2], tensors[3], latent_tokens_coord=tensors[0], xcoord=tensors[1])
hip = bench.hip_reference_pass(ts, model, tensors)       # the training path: grouped weight-gradient launch included
lat, x, p, t = [v.cpu() for v in tensors]
cfg = O.OracleConfig(radius=bench.RADIUS, hidden_size=64, lifting_channels=bench.C_LIFT, patch_size=bench.PATCH, tf_hidden_size=bench.HIDDEN,
                     latent_tokens_size=bench.LATENT, precompute_edges=True)
enc, dec = [O.radius_csr(x, lat, bench.RADIUS, exact=True)], [O.radius_csr(lat, x, bench.RADIUS, exact=True)]
t0 = time.time()
d = lambda v: v.double()
batch64 = dict(latent=d(lat), xcoord=d(x), pndata=d(p), target=d(t), encoder_nbrs=enc, decoder_nbrs=dec)
loss64, g64, _, _, pred64 = O.train_step({k: (d(v) if v.is_floating_point() else v) for k, v in sd.items()}, cfg, batch64, state=None, return_pred=True)
print(f"float64 oracle step {time.time() - t0:.1f} s; loss {float(loss64):.9f} vs HIP {hip['loss']:.9f}")
batch32 = dict(latent=lat, xcoord=x, pndata=p, target=t, encoder_nbrs=enc, decoder_nbrs=dec)
_, g32, _, _ = O.train_step(sd, cfg, batch32, state=None)
top = max(float(g.norm()) for g in g64.values())
rel = lambda a, b: float((a.double() - b.double()).norm()) / max(float(b.double().norm()), 1e-3 * top)
print(f"output rel-L2 vs float64: HIP {rel(hip['pred'], pred64):.2e}")
rows = sorted(((rel(hip["grads"][k], g), rel(g32[k], g), k) for k, g in g64.items()), reverse=True)
print("HIP vs f64   fp32-oracle vs f64   tensor")
for a, b, k in rows[:int(sys.argv[1]) if len(sys.argv) > 1 else 25]:
    print(f"  {a:.2e}     {b:.2e}          {k}")
