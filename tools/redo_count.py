#!/usr/bin/env python3
"""Which products of a training step (C2, or `c5`) send tiles through the fp16 pieces' second pass?  (gemm_split.hip "Dynamic range of the
fp16 pieces"; the device counter gaot_debug_split_redo_count), and how far below their tensor's maximum do the operands' rows and groups sit?
usage: redo_count.py [c5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaot_amd import ops, _lib
from gaot_amd.trainer import TrainStep

lib = _lib.load()
dev = torch.device("cuda:0")
torch.manual_seed(0)
if len(sys.argv) > 1 and sys.argv[1] == "c5":
    import tools.bench_configs as bc
    ts = bc.c5(build_only=True)
    ts.use_graph = False
else:
    model = bench.build_model().to(dev).train()
    lat, x, p, t = bench.synthetic(1234, dev)
    ts = TrainStep(model, use_graph=False)
    ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
for _ in range(int(os.environ.get("WARM", "2"))):
    ts.step()
torch.cuda.synchronize()
lib.gaot_debug_split_redo_count(1)
log = []
real_gemm, real_wgrad = ops.gemm, ops.wgrad_launch


def gemm(M, N, K, A, lda, a_kmajor, B, ldb, b_kmajor, out, ldc, **kw):
    r = real_gemm(M, N, K, A, lda, a_kmajor, B, ldb, b_kmajor, out, ldc, **kw)
    n = int(lib.gaot_debug_split_redo_count(1))
    kind = "tn" if not a_kmajor else ("nt" if b_kmajor else "nn")
    extra = ""
    if a_kmajor and K % 8 == 0 and A.dim() == 2 and A.shape[1] >= K:
        a = A[:, :K].abs()
        gm = a.reshape(M, K // 8, 8).amax(dim=2)
        nz = gm[gm > 0]
        rm = a.amax(dim=1)
        rnz = rm[rm > 0]
        q = torch.tensor([0.0, 0.001, 0.01, 0.1], device=a.device)
        extra = "  A: min group/max 2^%.1f, min row/max 2^%.1f, row quantiles(0,.1%%,1%%,10%%) 2^%s, zero rows %d" % (
            float(torch.log2(nz.min() / a.max())), float(torch.log2(rnz.min() / a.max())),
            [round(float(v), 1) for v in torch.log2(torch.quantile(rnz, q) / a.max())], int((rm == 0).sum()))
    log.append((kind, M, N, K, n, float(A.abs().max()), float(B.abs().max()), extra))
    return r


gemm.last_c_amax = None


def wgrad(items):
    r = real_wgrad(items)
    n = int(lib.gaot_debug_split_redo_count(1))
    log.append(("tn-grouped", len(items), 0, 0, n, 0.0, 0.0, ""))
    for it in items:
        g, x2 = it[0], it[2]
        # smallest non-zero 2 x 4 group maximum relative to the tensor's maximum, per operand
        def ratio(t_):
            a = t_.abs()
            r_, c_ = a.shape[0] // 2 * 2, a.shape[1] // 4 * 4
            gm = a[:r_, :c_].reshape(r_ // 2, 2, c_ // 4, 4).amax(dim=(1, 3))
            nz = gm[gm > 0]
            return float(nz.min() / a.max()) if nz.numel() else 1.0
        log.append(("   item", it[7], it[8], it[9], 0, ratio(g), ratio(x2), ""))
    return r


class G:
    def __call__(self, *a, **k):
        r = gemm(*a, **k)
        self.last_c_amax = real_gemm.last_c_amax
        return r


gg = G()
gg.last_c_amax = None
ops.gemm = gg
ops.wgrad_launch = wgrad
ts.step()
torch.cuda.synchronize()
for row in log:
    print("%-10s M=%-6d N=%-6d K=%-6d redo=%-5d  %.3e %.3e%s" % row)
print("total tiles through the second pass:", sum(r[4] for r in log))
