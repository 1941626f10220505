import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops, _lib as L
lib = L.load(); d = "cuda"
torch.manual_seed(0)
E, cin, n, act = 128, 4, 3, "relu"
x = torch.rand(E, cin, device=d) * 2 - 1
dims = [cin] + [64] * n
ws = [(torch.randn(dims[i + 1], dims[i], device=d) / dims[i] ** 0.5).requires_grad_() for i in range(n)]
bs = [(0.1 * torch.randn(64, device=d)).requires_grad_() for _ in range(n)]
acts = [act] * (n - 1) + ["none"]
dk = torch.randn(E, 64, device=d)
h = x.double()
wd = [w.detach().double().requires_grad_() for w in ws]; bd = [b.detach().double().requires_grad_() for b in bs]
for i in range(n):
    h = h @ wd[i].t() + bd[i]
    if i < n - 1: h = torch.relu(h)
gd = torch.autograd.grad(h, wd + bd, dk.double())
y = ops.mlp_chain(x, ws, bs, acts)
g = torch.autograd.grad(y, ws + bs, dk)
for i in range(n):
    a, b = g[n + i].double().cpu(), gd[n + i].cpu()
    print("db", i, "ratio first 8:", (a / b)[:8].numpy().round(3), " a:", a[:4].numpy().round(4), " b:", b[:4].numpy().round(4))
    print("   feature 32..35 ratio", (a / b)[32:36].numpy().round(3))
