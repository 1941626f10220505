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
LIB = _lib.load()
MARK = []          # WORDS index at which each step starts
_raw_wgrad = ops.wgrad_launch
def _wgrad(items):
    items2 = []
    for it in items:
        g, ldg, x2, ldx, out, ldo, cs, Mo, No, K = it[:10]
        ga, xa = (it[10], it[11]) if len(it) > 10 else (None, None)
        ga = ga if ga is not None else ops.amax_for(g)
        xa = xa if xa is not None else ops.amax_for(x2)
        items2.append((g, ldg, x2, ldx, out, ldo, cs, Mo, No, K, ga, xa))
    torch.cuda.synchronize()
    LIB.gaot_debug_split_redo_count(1)
    _raw_wgrad(items2)
    torch.cuda.synchronize()
    redone = int(LIB.gaot_debug_split_redo_count(1))
    # per item: shape, both magnitude words, both operands' true maxima and fp64 sums (are the INPUTS of the product the same?), the launch's redone tiles
    WORDS.append([(it[7], it[8], it[9], float(it[10].view(32, 32)[:, 0].max()), float(it[11].view(32, 32)[:, 0].max()),
                   float(it[0].abs().max()), float(it[2].abs().max()), float(it[0].double().sum()), float(it[2].double().sum()), redone) for it in items2])
ops.wgrad_launch = _wgrad
_raw_gemm = ops.gemm
def _gemm(M, N, K, A, lda, ak, Bm, ldb, bk, out, ldc, **kw):          # split-K products outside the grouped launch (small weight gradients)
    if kw.get("split_k", 0) <= 1:
        return _raw_gemm(M, N, K, A, lda, ak, Bm, ldb, bk, out, ldc, **kw)
    torch.cuda.synchronize(); LIB.gaot_debug_split_redo_count(1)
    r = _raw_gemm(M, N, K, A, lda, ak, Bm, ldb, bk, out, ldc, **kw)
    torch.cuda.synchronize()
    wa, wb = kw.get("a_amax"), kw.get("b_amax")
    WORDS.append([(M, N, K, -1.0 if wa is None else float(wa.view(32, 32)[:, 0].max()), -1.0 if wb is None else float(wb.view(32, 32)[:, 0].max()),
                   float(A.abs().max()), float(Bm.abs().max()), float(A.double().sum()), float(Bm.double().sum()),
                   "split_k %d" % kw["split_k"], int(LIB.gaot_debug_split_redo_count(1)))])
    return r
ops.gemm = _gemm
if seq:          # as det_batch: first training alone, recording its gradients per step; then the second against the record
    a, ma = make(False)
    rec = []
    for i in range(40):
        MARK.append(len(WORDS)); a.step(); torch.cuda.synchronize(); rec.append((a.bucket.flat.clone(), [q.detach().clone() for q in ma.parameters()]))
    words_a = list(WORDS); WORDS.clear(); mark_a = list(MARK) + [len(words_a)]
    b, mb = make(False)
    names = [n for n, _ in mb.named_parameters()]
    for i in range(40):
        w0 = len(WORDS); b.step(); torch.cuda.synchronize()
        if not torch.equal(b.bucket.flat, rec[i][0]):
            d = (b.bucket.flat - rec[i][0]).abs()
            print("step", i, "gradients differ in", int((d > 0).sum()), "elements, max abs", float(d.max()), "of max", float(rec[i][0].abs().max()), flush=True)
            la, lb = words_a[mark_a[i]:mark_a[i + 1]], WORDS[w0:]
            print("    dW launches in this step:", len(la), len(lb), " redone tiles per launch:", [l[0][-1] for l in la], [l[0][-1] for l in lb], flush=True)
            for li, (xa_, xb_) in enumerate(zip(la, lb)):
                for j, (wa_, wb_) in enumerate(zip(xa_, xb_)):
                    if wa_ != wb_:
                        print("    launch", li, "item", j, "M N K", wa_[:3], "first run", wa_[3:], "second run", wb_[3:], flush=True)
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
