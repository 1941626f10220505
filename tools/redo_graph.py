#!/usr/bin/env python3
"""tiles through the second pass during hipGraph-replayed C2 steps (the bench's own path)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
import bench
from gaot_amd import _lib
from gaot_amd.trainer import TrainStep
lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = bench.build_model().to(dev).train()
lat, x, p, t = bench.synthetic(1234, dev)
ts = TrainStep(model, use_graph=True)
ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
for _ in range(5):
    ts.step()
torch.cuda.synchronize()
print("after warm-up:", lib.gaot_debug_split_redo_count(1))
t0 = time.perf_counter()
for _ in range(20):
    ts.step()
torch.cuda.synchronize()
print("20 replayed steps: redo tiles", lib.gaot_debug_split_redo_count(1), "ms/step", (time.perf_counter() - t0) / 20 * 1e3)
