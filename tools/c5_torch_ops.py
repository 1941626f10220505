#!/usr/bin/env python3
"""Which launches of a C5 (3-D cloud) training step are NOT gaot kernels, and where they come from: one eager step under torch.profiler
with stacks (ATen ops that launch a kernel), plus the fallback absmax launches' call sites (GAOT_AMAX_TRACE=1)."""
import os, sys
os.environ["GAOT_AMAX_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools.bench_configs as bc
from torch.profiler import profile, ProfilerActivity

ts = bc.c5(build_only=True)
ts.use_graph = False
for _ in range(3):
    ts.step()
torch.cuda.synchronize()
print("==== traced step", flush=True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    ts.step()
    torch.cuda.synchronize()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.device_time_total > 0 and ev.name not in ("aten::empty", "aten::view"):
        st = [s for s in (ev.stack or []) if "gaot_amd" in s or "bench" in s][:4]
        print(f"{ev.name:28s} {ev.device_time_total:7.1f} us  shapes {ev.input_shapes if ev.input_shapes else ''}  <- {' < '.join(s.split('/')[-1] for s in st)}")
