#!/usr/bin/env python3
import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
import torch
from tests.test_configs_gpu import _c3_case, csr_dict, dev
from gaot_amd.trainer import TrainStep
from gaot_amd import plan as P
mode = sys.argv[1]
vx = sys.argv[2] == "vx"
graph = sys.argv[3] == "graph"
kw_s = dict(sampling_strategy="max_neighbors", max_neighbors=6) if mode == "maxn" else dict(sampling_strategy="ratio", sample_ratio=0.6)
model, sd, ocfg, lat, x, p, tgt, enc, dec = _c3_case(B=2, N=4096, spread=0.2, seed=3, **kw_s)
if not vx:
    x = x[0]
    enc, dec = [enc[0][0]], [dec[0][0]]
model.to(dev()).train()
todev = (lambda rows: [[csr_dict(c) for c in row] for row in rows]) if vx else (lambda rows: [csr_dict(c) for c in rows])
kw = dict(latent_tokens_coord=lat.to(dev()), xcoord=x.to(dev()), encoder_nbrs=todev(enc), decoder_nbrs=todev(dec))
if len(sys.argv) > 4:
    from gaot_amd import ops
    if "rec" in sys.argv[4]:
        P.DROP_RECORD = []
    pred = model(pndata=p.to(dev()), **kw)
    loss = ops.mse_loss(pred, tgt.to(dev()))
    loss.backward()
    torch.cuda.synchronize()
    print("plain pass done", float(loss), len(P.DROP_RECORD or []), flush=True)
    if "second" in sys.argv[4]:
        model(pndata=p.to(dev()), **kw)
    P.DROP_RECORD = None
ts = TrainStep(model, lr=0.0, weight_decay=0.0, use_graph=graph)
ts.bind(p.to(dev()), tgt.to(dev()), **kw)
for i in range(4):
    l = ts.step()
    torch.cuda.synchronize()
    print(mode, vx, graph, i, float(l), flush=True)
