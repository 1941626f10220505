#!/usr/bin/env python3
"""How much idle capacity do the step's kernels leave?  Two INDEPENDENT TrainSteps (two models, the bench configuration) replayed on two
streams at once against the same two replayed back to back: if the pair finishes much sooner, overlapping independent work inside one
step (the deferred weight gradients next to the input-gradient chain) has something to harvest."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaot_amd.trainer import TrainStep
from gaot_amd import ops
dev = torch.device("cuda:0")
steps = []
for seed in (0, 1):
    ops.register_grad_slots([], [])
    torch.manual_seed(seed)
    m = bench.build_model().to(dev).train()
    lat, x, p, t = bench.synthetic(1234 + seed, dev)
    ts = TrainStep(m, lr=8e-4, weight_decay=1e-5, use_graph=True)
    ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
    for _ in range(3): ts.step()
    steps.append(ts)
torch.cuda.synchronize()
def replay(ts):
    ts._graphs[0].replay()
    if ts._g_opt is not None:          # (one rank: the optimizer is inside the step's graph)
        ts._g_opt.replay()
N = 100
t0 = time.perf_counter()
for _ in range(N):
    replay(steps[0]); replay(steps[1])
torch.cuda.synchronize()
seq = (time.perf_counter() - t0) / N
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    with torch.cuda.stream(s1): replay(steps[0])
    with torch.cuda.stream(s2): replay(steps[1])
torch.cuda.synchronize()
par = (time.perf_counter() - t0) / N
print(f"two steps back to back: {seq * 1e3:.3f} ms; on two streams: {par * 1e3:.3f} ms ({seq / par:.3f}x)")
