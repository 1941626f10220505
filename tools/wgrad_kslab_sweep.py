import sys, os
sys.path.insert(0, "/root/repo")
import torch
from gaot_amd import ops, _lib
lib = _lib.load(); dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
shapes = [(2048, 256), (256, 1024), (256, 256), (768, 256)] * 3 + [(256, 256), (256, 512)]
g = torch.Generator().manual_seed(0)
ops_in = [(torch.randn(T, Mo, generator=g).to(dev), torch.randn(T, No, generator=g).to(dev), torch.empty(Mo, No, device=dev)) for Mo, No in shapes]
def grouped(items):
    with ops.deferred_wgrad():
        for dy, x, out in items: ops.matmul_tn(dy, x, out=out, final=True)
def timed(fn, n=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(20e-3 * 2.0e9)); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, items in (("all 14", ops_in), ("one phase (4)", ops_in[:4])):
    for ks in (0, 4096, 2736, 2048, 1376, 1024, 512):
        lib.gaot_debug_set_wgrad_kslab(ks)
        print(f"{name}: kslab {ks}: {timed(lambda: grouped(items)):.1f} us", flush=True)
lib.gaot_debug_set_wgrad_kslab(0)
