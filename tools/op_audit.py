#!/usr/bin/env python3
"""Which torch (non-gaot) kernels does one eager training step still launch, and from where?
Prints aten ops that launched device kernels, grouped by op + input shapes, with the python call site."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from gaot_amd.trainer import TrainStep

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench.build_model().to(dev).train()
lat, x, p, t = bench.synthetic(1234, dev)
ts = TrainStep(model, lr=8e-4, weight_decay=1e-5, use_graph=False)
ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
for _ in range(3):
    ts.step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    ts.step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
    if e.device_time_total > 0 and e.key.startswith("aten::") and e.self_device_time_total > 0:
        stack = [s for s in e.stack if "gaot_amd" in s or "bench.py" in s]
        rows.append((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:90], " <- ".join(x.split("/")[-1][:60] for x in stack[:3]) if stack else str(e.stack[:2])[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"torch-launched kernel time in one eager step: {tot:.0f} us")
for us, n, key, shp, st in rows[:45]:
    print(f"{us:8.1f}us x{n:<3d} {key:28s} {shp:90s} {st}")
