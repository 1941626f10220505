#!/usr/bin/env python3
"""Where the host time of the reference's vx loop under autograph goes: cProfile over 20 shuffled steps (C3 shapes)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace as NS
import torch
import bench
from gaot_amd.model.gaot import GAOT
from gaot_amd.model.layers.magno import MAGNOConfig
from gaot_amd.model.layers.attn import TransformerConfig
from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch
from tests._workloads import grid, naca_points

dev = torch.device("cuda:0")
B, N, NSD = 16, 8192, 32
mc = MAGNOConfig(radius=bench.RADIUS, lifting_channels=bench.C_LIFT, precompute_edges=True)
torch.manual_seed(0)
m2 = GAOT(3, 1, NS(args=NS(magno=mc, transformer=TransformerConfig(patch_size=bench.PATCH, hidden_size=bench.HIDDEN)), latent_tokens_size=bench.LATENT)).to(dev).train()
g = torch.Generator().manual_seed(0)
lat = grid(bench.LATENT)
x_all = torch.stack([naca_points(N, g, 0.15) for i in range(NSD)])
p_all, t_all = torch.randn(NSD, N, 3, generator=g), torch.randn(NSD, N, 1, generator=g)
ns = NeighborSearch("auto")
xd_all, latd = x_all.to(dev), lat.to(dev)
enc_all = [[ns(xd_all[i], latd, bench.RADIUS)] for i in range(NSD)]
dec_all = [[ns(latd, xd_all[i], bench.RADIUS)] for i in range(NSD)]
opt = torch.optim.AdamW(m2.parameters(), lr=8e-4, weight_decay=1e-5)
lossf = torch.nn.MSELoss()
gsh = torch.Generator().manual_seed(7)
draw = lambda: torch.randperm(NSD, generator=gsh)[:B].tolist()


def loop_step(b):
    xb, yb = p_all[b].to(dev), t_all[b].to(dev)
    xc = x_all[b].to(dev)
    opt.zero_grad()
    out_ = m2(latent_tokens_coord=latd, xcoord=xc, pndata=xb, encoder_nbrs=[enc_all[i] for i in b], decoder_nbrs=[dec_all[i] for i in b])
    lossf(out_, yb).backward()
    opt.step()
    return type(out_.grad_fn).__name__ == "_GraphedStepBackward"


for _ in range(8):
    loop_step(draw())
torch.cuda.synchronize()
t0 = time.perf_counter()
n = sum(loop_step(draw()) for _ in range(20))
torch.cuda.synchronize()
print(f"{1e3 * (time.perf_counter() - t0) / 20:.2f} ms/step, graphed {n}/20")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    loop_step(draw())
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(35)
