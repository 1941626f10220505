import os, sys
sys.path.insert(0, "/root/repo")
import torch
from gaot_amd import _lib
import tools.bench_configs as bc
lib = _lib.load()
ts = bc.c5(build_only=True)
ts.use_graph = False
for i in range(12):
    ts.step(); torch.cuda.synchronize()
    print("eager step", i, "redo tiles", lib.gaot_debug_split_redo_count(1))
ts2 = bc.c5(build_only=True)
for i in range(8):
    ts2.step(); torch.cuda.synchronize()
    print("graph step", i, "redo tiles", lib.gaot_debug_split_redo_count(1))
