#!/usr/bin/env python3
"""tools/det_batch.py's history (two graph trainings of 100 steps), then two EAGER trainings from the same seed stepped in LOCKSTEP: at the first
step whose flat gradients differ, which parameters' gradients are they, by how much?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gaot_amd import ops, _lib
from gaot_amd.trainer import TrainStep
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
def make(graph):
    ops.register_grad_slots([], [])
    torch.manual_seed(0)
    model = bench.build_model().to(dev).train()
    lat, x, p, t = bench.synthetic(1234, dev)
    p, t = p[:B].contiguous(), t[:B].contiguous()
    ts = TrainStep(model, lr=8e-4, weight_decay=1e-5, use_graph=graph)
    ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
    return ts, model
for rep in range(2):
    ts, model = make(True)
    for i in range(100): ts.step()
    torch.cuda.synchronize()
    del ts, model
seq = "--seq" in sys.argv
WORDS = []
_raw_wgrad = ops.wgrad_launch
def _wgrad(items):
    items2 = []
    for it in items:
        g, ldg, x2, ldx, out, ldo, cs, Mo, No, K = it[:10]
        ga, xa = (it[10], it[11]) if len(it) > 10 else (None, None)
        ga = ga if ga is not None else ops.amax_for(g)
        xa = xa if xa is not None else ops.amax_for(x2)
        items2.append((g, ldg, x2, ldx, out, ldo, cs, Mo, No, K, ga, xa))
    _raw_wgrad(items2)
    torch.cuda.synchronize()
    WORDS.append([(it[7], it[8], it[9], float(it[10].view(32, 32)[:, 0].max()), float(it[11].view(32, 32)[:, 0].max()),
                   float(it[0].abs().max()), float(it[2].abs().max())) for it in items2])
ops.wgrad_launch = _wgrad
if seq:          # as det_batch: first training alone, recording its gradients per step; then the second against the record
    a, ma = make(False)
    rec = []
    for i in range(40):
        a.step(); torch.cuda.synchronize(); rec.append((a.bucket.flat.clone(), [q.detach().clone() for q in ma.parameters()]))
    words_a = list(WORDS); WORDS.clear()
    b, mb = make(False)
    names = [n for n, _ in mb.named_parameters()]
    for i in range(40):
        b.step(); torch.cuda.synchronize()
        if not torch.equal(b.bucket.flat, rec[i][0]):
            d = (b.bucket.flat - rec[i][0]).abs()
            print("step", i, "gradients differ in", int((d > 0).sum()), "elements, max abs", float(d.max()), "of max", float(rec[i][0].abs().max()), flush=True)
            for j, (wa_, wb_) in enumerate(zip(words_a[i], WORDS[i])):
                if wa_ != wb_:
                    print("    dW item", j, "M N K", wa_[:3], "words (dY, X) first run", wa_[3:5], "second run", wb_[3:5], "true max |dY|, |X|", wa_[5:], wb_[5:], flush=True)
            pid = {id(q): n for n, q in mb.named_parameters()}
            for q, o in zip(b.bucket.params, b.bucket.offsets):
                dd = d[o:o + q.numel()]
                if float(dd.max()) > 0:
                    ref = rec[i][0][o:o + q.numel()]
                    print("   ", pid[id(q)], tuple(q.shape), "differing", int((dd > 0).sum()), "max abs", float(dd.max()), "grad max", float(ref.abs().max()), flush=True)
            break
    else:
        print("no difference in 40 steps")
else:
    a, ma = make(False); b, mb = make(False)
    for i in range(60):
        a.step(); b.step(); torch.cuda.synchronize()
        if not torch.equal(a.bucket.flat, b.bucket.flat):
            d = (a.bucket.flat - b.bucket.flat).abs()
            print("lockstep step", i, "gradients differ in", int((d > 0).sum()), "elements, max abs", float(d.max()), flush=True)
            break
    else:
        print("lockstep: no difference in 60 steps")
