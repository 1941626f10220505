#!/usr/bin/env python3
"""A few 10-step autoregressive rollouts of the C4 shape (4 x 16 384 nodes, stepper 'time_der') for the kernel tracer:
    rocprofv3 --kernel-trace -d OUT -o r -- python tools/rollout_trace.py ; python tools/trace_tail.py OUT/*.db 400"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
import numpy as np, torch, time
from gaot_amd.model.gaot import GAOT
from gaot_amd.model.layers.magno import MAGNOConfig
from gaot_amd.model.layers.attn import TransformerConfig
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, N = 4, 16384
mc = MAGNOConfig(radius=0.033, lifting_channels=64)
model = GAOT(4, 2, NS(args=NS(magno=mc, transformer=TransformerConfig(patch_size=2, hidden_size=256)), latent_tokens_size=[64, 64])).to(dev).eval()
axes = [torch.linspace(-1, 1, n) for n in (64, 64)]
lat = torch.stack(torch.meshgrid(*axes, indexing="ij"), -1).reshape(-1, 2).to(dev)
x = (torch.rand(N, 2, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
xb = torch.randn(B, N, 4, device=dev)
stats = {"u": {"mean": torch.zeros(2), "std": torch.ones(2)}, "der": {"mean": torch.zeros(2), "std": torch.ones(2)},
         "start_time": {"mean": 0.0, "std": 1.0}, "time_diffs": {"mean": 0.0, "std": 1.0}}
tv, ti = np.linspace(0, 1, 21), np.arange(0, 22, 2)[:11]
roll = lambda: model.autoregressive_predict(x_batch=xb[..., :2], time_indices=ti, t_values=tv, stats=stats, stepper_mode="time_der", latent_tokens_coord=lat, fixed_coord=x)
for _ in range(3):
    roll()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    roll()
torch.cuda.synchronize()
print("rollout ms", (time.perf_counter() - t0) / 5 * 1e3)
