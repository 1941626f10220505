"""The 13 weight-gradient products of one training step at the bench configuration (3 transformer blocks: dW13, dW2, dWo, dWqkv;
patch_linear; skip_proj), one gaot_gemm_f32 (+ split-K reduce) each vs ONE gaot_gemm_tn_grouped launch.  Prints us per step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops

dev = torch.device("cuda:0")
T = 8192
shapes = [(2048, 256), (256, 1024), (256, 256), (768, 256)] * 3 + [(256, 256), (256, 512)]
g = torch.Generator().manual_seed(0)
ops_in = []
for Mo, No in shapes:
    ops_in.append((torch.randn(T, Mo, generator=g).to(dev), torch.randn(T, No, generator=g).to(dev), torch.empty(Mo, No, device=dev)))
flops = sum(2.0 * Mo * No * T for Mo, No in shapes)


def single():
    for dy, x, out in ops_in:
        ops.matmul_tn(dy, x, out=out)


def grouped():
    with ops.deferred_wgrad():
        for dy, x, out in ops_in:
            ops.matmul_tn(dy, x, out=out, final=True)


def timed(fn, n=30):
    for _ in range(3):
        fn()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(gr):
        fn()
    gr.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        gr.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


single()
ref = [o.clone() for _, _, o in ops_in]
from gaot_amd import _lib
lib = _lib.load()
us_s = timed(single)
for rep in range(2):
    for bm in (128, 256):
        for kslab in (2048, 2752, 4096, 8192):
            lib.gaot_debug_set_wgrad_tile_rows(bm)
            lib.gaot_debug_set_wgrad_kslab(kslab)
            for _, _, o in ops_in:
                o.zero_()
            grouped()
            torch.cuda.synchronize()
            err = max(float((o.double() - r.double()).norm() / r.double().norm()) for (_, _, o), r in zip(ops_in, ref))
            us_g = timed(grouped)
            print(f"tile rows {bm} kslab {kslab}: single launches {us_s:8.1f} us ({flops / us_s / 1e6:6.1f} TF/s)   grouped {us_g:8.1f} us ({flops / us_g / 1e6:6.1f} TF/s)   "
                  f"max rel diff vs single launches {err:.2e}", flush=True)
lib.gaot_debug_set_wgrad_tile_rows(128)
lib.gaot_debug_set_wgrad_kslab(0)
