"""Grouped weight-gradient launch at the bench shapes: exact three-piece operands against two rounded pieces -- time and error vs float64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda:0")
T = 8192
shapes = [(2048, 256), (256, 1024), (256, 256), (768, 256)] * 3 + [(256, 256), (256, 512)]
g = torch.Generator().manual_seed(0)
ops_in = [(torch.randn(T, Mo, generator=g).to(dev), torch.randn(T, No, generator=g).to(dev), torch.empty(Mo, No, device=dev)) for Mo, No in shapes]
ref = [(dy.double().t() @ x.double()) for dy, x, _ in ops_in]
def grouped():
    with ops.deferred_wgrad():
        for dy, x, out in ops_in: ops.matmul_tn(dy, x, out=out, final=True)
def timed(fn, n=40):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(20e-3 * 2.0e9)); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for rep in range(2):
    for pieces in (3, 2):
        ops.set_gemm_pieces(pieces)
        grouped(); torch.cuda.synchronize()
        errs = [float((o.double() - r).norm() / r.norm()) for (_, _, o), r in zip(ops_in, ref)]
        print(f"pieces {pieces}: {timed(grouped):.1f} us   error vs float64: max {max(errs):.2e}  mean {sum(errs)/len(errs):.2e}", flush=True)
ops.set_gemm_pieces(2)
