import sys
sys.path.insert(0, "/root/repo")
import torch
from gaot_amd import ops
dev = "cuda"
torch.manual_seed(0)
B = 4
qkv = torch.randn(B, 1024, 768, device=dev, requires_grad=True)
go = torch.randn(B, 1024, 256, device=dev)
ref = None
nbad = 0
import random
random.seed(1)
for it in range(60):
    # poison the allocator's free blocks: whatever the op reads without having written it shows up as NaN
    junk = [torch.full((random.randint(1, 64) * 65536,), float('nan'), device=dev) for _ in range(12)]
    del junk
    ops.begin_pass()
    o = ops.attention(qkv, 8, 8, 32)
    g, = torch.autograd.grad(o, qkv, go)
    torch.cuda.synchronize()
    if ref is None: ref = g.clone()
    elif not torch.equal(g, ref):
        d = (g - ref).abs()
        nbad += 1
        blk = ["q", "k", "v"]
        per = [float(d[..., i*256:(i+1)*256].max()) for i in range(3)]
        cnt = int((d > 0).sum())
        print("iter", it, "differs: elements", cnt, "max diff per block q/k/v", per, "ref max", float(ref.abs().max()), flush=True)
print("runs differing from the first:", nbad)
