#!/usr/bin/env python3
"""Same-box A/B of the bench step (C2, TrainStep under hipGraph) for an integer debug switch of the library (or `ops.<NAME>`, a module-level
switch of gaot_amd.ops):
    python tools/step_ab.py gaot_debug_set_gemm_ad 0 1 [2 ...]
builds one TrainStep per value (the graph is captured with the switch set), then times them in alternation (3 rounds x 60 steps each)
and prints ms per step per value.  --c4: the 4 096-token batch of BASELINE configs[3] instead."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaot_amd import ops, _lib
from gaot_amd.trainer import TrainStep

lib = _lib.load()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
if args[0].startswith("ops."):          # a module-level switch of gaot_amd.ops instead of a library function: python tools/step_ab.py ops._NARROW_TILES_1K 0 1
    _name = args[0][4:]
    def fn(v):
        old = getattr(ops, _name); setattr(ops, _name, v); return old
else:
    fn = getattr(lib, args[0])
values = [int(v) for v in args[1:]]
dev = torch.device("cuda:0")
steps = []
for v in values:
    old = fn(v)
    ops._PATH_CACHE.clear()          # (the dry-run answers of the dispatcher are cached per shape)
    ops.register_grad_slots([], [])
    torch.manual_seed(0)
    model = bench.build_model().to(dev).train()
    lat, x, p, t = bench.synthetic(1234, dev)
    if "--c4" in sys.argv:
        p, t = p[:4].contiguous(), t[:4].contiguous()
    ts = TrainStep(model, lr=8e-4, weight_decay=1e-5, use_graph=True)
    ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
    for _ in range(8):
        ts.step()
    torch.cuda.synchronize()
    steps.append((v, ts, model))
    fn(old)
res = {v: [] for v in values}
redo = {}
for v, ts, _ in steps:          # tiles that take the fp16 pieces' second pass in ONE step of each variant (benign data: none expected)
    lib.gaot_debug_split_redo_count(1)
    ts.step(); torch.cuda.synchronize()
    redo[v] = int(lib.gaot_debug_split_redo_count(1))
for rnd in range(3):
    for v, ts, _ in steps:
        for _ in range(5):
            ts.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(60):
            ts.step()
        torch.cuda.synchronize()
        res[v].append((time.perf_counter() - t0) / 60 * 1e3)
for v in values:
    print(f"{args[0]}({v}): " + "  ".join(f"{r:.4f}" for r in res[v]) + f"   best {min(res[v]):.4f} ms/step   loss {float(steps[values.index(v)][1]._loss):.6e}   redone tiles per step {redo[v]}", flush=True)
