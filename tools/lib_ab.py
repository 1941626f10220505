#!/usr/bin/env python3
"""Same-box A/B of two BUILDS of the library on the bench step (C2; --c4 / --c5 / --c3: the other configurations):
    python tools/lib_ab.py gaot_amd/lib/libgaot_hip_base.so gaot_amd/lib/libgaot_hip.so [--c4]
runs one child process per build and round (GAOT_HIP_LIB), alternating, 3 rounds; each child captures the step, times 3 x 100 replays and
prints ms per step, the loss after its last step and a checksum of the parameters (bit-identical builds print identical ones)."""
import json, os, subprocess, sys, time

if os.environ.get("GAOT_LIB_AB_CHILD"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    which = os.environ["GAOT_LIB_AB_CHILD"]
    if which in ("c2", "c4"):
        import bench
        from gaot_amd.trainer import TrainStep
        torch.manual_seed(0)
        dev = torch.device("cuda:0")
        model = bench.build_model().to(dev).train()
        lat, x, p, t = bench.synthetic(1234, dev)
        if which == "c4":
            p, t = p[:4].contiguous(), t[:4].contiguous()
        ts = TrainStep(model, lr=8e-4, weight_decay=1e-5, use_graph=True)
        ts.bind(p, t, latent_tokens_coord=lat, xcoord=x)
    else:
        import tools.bench_configs as bc
        if which == "c5":
            ts = bc.c5(build_only=True)
        else:
            raise SystemExit("child: c2 | c4 | c5")
        model = ts.model if hasattr(ts, "model") else None
    for _ in range(10):
        ts.step()
    torch.cuda.synchronize()
    res = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(100):
            ts.step()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) * 10)
    cs = 0.0
    if model is not None:
        cs = float(sum(q.detach().double().sum() for q in model.parameters()))
    print(json.dumps({"ms": res, "loss": float(ts._loss), "param_sum": cs}))
    sys.exit(0)

libs = [a for a in sys.argv[1:] if not a.startswith("--")]
which = ([a[2:] for a in sys.argv[1:] if a.startswith("--")] or ["c2"])[0]
out = {l: [] for l in libs}
for rnd in range(3):
    for l in (libs if rnd % 2 == 0 else libs[::-1]):
        env = dict(os.environ, GAOT_HIP_LIB=os.path.abspath(l), GAOT_LIB_AB_CHILD=which)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        line = [x for x in r.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(l, "FAILED", r.stderr[-500:])
            continue
        out[l].append(json.loads(line[-1]))
for l in libs:
    ms = [m for d in out[l] for m in d["ms"]]
    print(f"{l}: best {min(ms):.4f}  all {' '.join(f'{m:.4f}' for m in ms)}  loss {out[l][-1]['loss']:.9e}  param_sum {out[l][-1]['param_sum']:.12e}", flush=True)
