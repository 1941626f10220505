"""Long-run soak of the replayed training paths (not part of the product; `gpurun -- python tools/soak.py [steps]`).
Three loops of `steps` steps each, every one watched for device-memory growth after its captures have settled, for a non-finite loss
and for a growing number of captured graph sets:
  vx-trainstep   TrainStep.step(pndata, target, xcoord=, encoder_nbrs=, decoder_nbrs=) under a shuffling loader (a new composition of 8 of 32
                 resident meshes every step: plan.StaticUnion re-composed on the device inside the captured step)
  vx-reference   the unchanged reference loop (zero_grad / model(...) / MSELoss / backward / torch.optim.AdamW) over the same loader (autograph)
  fx-reference   the unchanged reference loop on one fixed mesh with a new field batch every step (autograph, patch-major latent grid)
Prints one JSON line per loop; exit code 1 if any check fails."""
import json
import os
import sys
import time
from types import SimpleNamespace as NS

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaot_amd.model.gaot import GAOT                                            # noqa: E402
from gaot_amd.model.layers.attn import TransformerConfig                        # noqa: E402
from gaot_amd.model.layers.magno import MAGNOConfig                             # noqa: E402
from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch         # noqa: E402
from gaot_amd.trainer import TrainStep                                          # noqa: E402
from tests._workloads import grid, naca_points                                  # noqa: E402

LATENT, RADIUS, N, B, NS_DATA = [32, 32], 0.066, 2048, 8, 32


def mem():
    torch.cuda.synchronize()
    return torch.cuda.memory_allocated(), torch.cuda.memory_reserved()


def watch(name, steps, step_fn, extra):
    """run `steps` steps; memory is sampled at 1/4, 1/2, 3/4 and the end: allocated bytes must not grow after the first quarter"""
    marks, losses = [], []
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step_fn(i)
        if (i + 1) % max(steps // 4, 1) == 0:
            losses.append(float(loss))
            marks.append(mem() + (extra(),))
    dt = time.perf_counter() - t0
    ok = all(l == l and abs(l) < 1e30 for l in losses) and marks[-1][0] <= marks[0][0] + (1 << 20) and marks[-1][2] == marks[0][2]
    print(json.dumps({"loop": name, "steps": steps, "ms_per_step": 1e3 * dt / steps, "loss_at_quarters": losses,
                      "allocated_MB_at_quarters": [m[0] / 2 ** 20 for m in marks], "reserved_MB_at_quarters": [m[1] / 2 ** 20 for m in marks],
                      "graph_sets_at_quarters": [m[2] for m in marks], "ok": ok}), flush=True)
    return ok


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(0)
    lat = grid(LATENT).to(dev)
    x_all = torch.stack([naca_points(N, g, 0.12 + 0.06 * (i % 8) / 7) for i in range(NS_DATA)]).to(dev)
    p_all, t_all = torch.randn(NS_DATA, N, 3, generator=g).to(dev), torch.randn(NS_DATA, N, 1, generator=g).to(dev)
    ns = NeighborSearch("auto")
    enc_all = [[ns(x_all[i], lat, RADIUS)] for i in range(NS_DATA)]
    dec_all = [[ns(lat, x_all[i], RADIUS)] for i in range(NS_DATA)]
    mk = lambda pre: GAOT(3, 1, NS(args=NS(magno=MAGNOConfig(radius=RADIUS, lifting_channels=64, precompute_edges=pre),
                                           transformer=TransformerConfig(patch_size=2, hidden_size=256)), latent_tokens_size=LATENT)).to(dev).train()
    gsh = torch.Generator().manual_seed(7)
    draw = lambda: torch.randperm(NS_DATA, generator=gsh)[:B].tolist()
    pick = lambda rows, b: [rows[i] for i in b]
    ok = True

    # ---- vx, the repo's own harness
    model = mk(True)
    ts = TrainStep(model, lr=1e-4, weight_decay=1e-5, use_graph=True)
    b0 = draw()
    ix = torch.tensor(b0, device=dev)
    ts.bind(p_all[ix], t_all[ix], latent_tokens_coord=lat, xcoord=x_all[ix], encoder_nbrs=pick(enc_all, b0), decoder_nbrs=pick(dec_all, b0))

    def vx_ts(i):
        b = draw()
        ix = torch.tensor(b, device=dev)
        return ts.step(p_all[ix], t_all[ix], xcoord=x_all[ix], encoder_nbrs=pick(enc_all, b), decoder_nbrs=pick(dec_all, b))
    for i in range(200):          # captures settle (one per edge bucket the loader lands in)
        vx_ts(i)
    ok &= watch("vx-trainstep", steps, vx_ts, lambda: len(ts._graph_sets))
    del ts, model
    torch.cuda.empty_cache()

    # ---- vx, the unchanged reference loop
    model = mk(True)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
    lossf = torch.nn.MSELoss()

    def vx_ref(i):
        b = draw()
        ix = torch.tensor(b, device=dev)
        opt.zero_grad()
        pred = model(latent_tokens_coord=lat, xcoord=x_all[ix], pndata=p_all[ix], encoder_nbrs=pick(enc_all, b), decoder_nbrs=pick(dec_all, b))
        loss = lossf(pred, t_all[ix])
        loss.backward()
        opt.step()
        return loss
    for i in range(200):
        vx_ref(i)
    ok &= watch("vx-reference", steps, vx_ref, lambda: 0)
    del opt, model
    torch.cuda.empty_cache()

    # ---- fx, the unchanged reference loop (module-owned graphs)
    model = mk(False)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5)
    x0 = x_all[0].contiguous()

    def fx_ref(i):
        b = draw()
        ix = torch.tensor(b, device=dev)
        opt.zero_grad()
        pred = model(latent_tokens_coord=lat, xcoord=x0, pndata=p_all[ix])
        loss = lossf(pred, t_all[ix])
        loss.backward()
        opt.step()
        return loss
    for i in range(100):
        fx_ref(i)
    ok &= watch("fx-reference", steps, fx_ref, lambda: 0)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
