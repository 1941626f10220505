import sys, os
sys.path.insert(0, "/root/repo")
import torch
from gaot_amd import ops, _lib
lib = _lib.load()
dev = "cuda"
g = torch.Generator().manual_seed(0)
bad = 0
for it in range(150):
    M = [8192, 4096, 8192, 1000][it % 4]; N = 256; K = [256, 256, 128, 96][it % 4]
    x = torch.randn(M, K, generator=g).cuda(); dy = torch.randn(M, N, generator=g).cuda()
    w = torch.nn.Parameter((torch.randn(N, K, generator=g) * 0.06).cuda())
    w2 = torch.nn.Parameter((torch.randn(K, N, generator=g) * 0.06).cuda())
    outs = []
    for mode in (0, 3, 3):
        old = lib.gaot_debug_set_gemm_ad_narrow(mode)
        ops._PATH_CACHE.clear()
        ops.begin_pass(); ops.refresh_weight_amax([w, w2])
        with torch.no_grad():
            a = ops.linear_nt(x, w.detach()); b = ops.matmul_nn(dy, w.detach()) if N == 256 else None
            c = ops.linear_nt(a, w2.detach())
        torch.cuda.synchronize()
        outs.append((a.clone(), b.clone(), c.clone()))
        lib.gaot_debug_set_gemm_ad_narrow(old)
    for i, nm in enumerate(("nt", "nn", "nt2")):
        if not torch.equal(outs[1][i], outs[2][i]):
            bad += 1; print("NONDETERMINISTIC", it, M, K, nm, float((outs[1][i] - outs[2][i]).abs().max()))
        if lib.gaot_debug_last_gemm_path() == 3 and M >= 8000 and not torch.equal(outs[0][i], outs[1][i]):
            bad += 1; print("64x64 != 64x128", it, M, K, nm, float((outs[0][i] - outs[1][i]).abs().max()))
print("done, mismatches:", bad)
