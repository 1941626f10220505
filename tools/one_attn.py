#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaot_amd import ops
dev = torch.device("cuda:0")
qkv = torch.randn(8, 1024, 768, device=dev, requires_grad=True)
go = torch.randn(8, 1024, 256, device=dev)
for _ in range(6):
    o = ops.attention(qkv, 8, 8, 32)
    o.backward(go)
torch.cuda.synchronize()
