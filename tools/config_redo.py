#!/usr/bin/env python3
"""tiles through the fp16 pieces' second pass per hipGraph-replayed step of C5 (after the warm-up), and the step time"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import _lib
import tools.bench_configs as bc
lib = _lib.load()
ts = bc.c5(build_only=True)
for _ in range(5):
    ts.step()
torch.cuda.synchronize()
print("warm-up (eager + capture + 3 replays):", lib.gaot_debug_split_redo_count(1), "tiles")
t0 = time.perf_counter()
for _ in range(10):
    ts.step()
torch.cuda.synchronize()
print("10 replayed steps:", lib.gaot_debug_split_redo_count(1), "tiles;", (time.perf_counter() - t0) * 100, "ms per step")
