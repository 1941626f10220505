#!/usr/bin/env python3
"""Generate tests/golden/fuzz_reference_steps.npz: ONE trainer step (forward, MSELoss, backward) of the *imported reference's* GAOT on the
first 24 configurations of the randomised parity sweep (tools/fuzz_parity.py: `draw`, `make_batch`), with the weights oracle.make_state_dict
lays out for the configuration (seeded; loaded strictly into the reference model).  Recorded per seed: the prediction, the loss, every
parameter gradient's norm and its projection on a fixed cosine vector -- data, not source.  tests/test_fuzz_cpu.py holds the ORACLE to it:
the checker of the sweep is itself checked against the reference across the sweep's option space, not only on the 24 named fixtures.
Runs ONLY in the build container (needs /root/reference; the stand-ins of make_golden.py for the packages the image lacks).

Usage:  python tests/golden/make_fuzz_reference_steps.py"""
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as MG                                    # noqa: E402  (install_standins only)
from oracle import gaot_oracle as O                         # noqa: E402  (make_state_dict: the weights both sides load)
from tools import fuzz_parity as F                          # noqa: E402

N_SEEDS = 24


def proj(g):
    v = g.detach().double().reshape(-1)
    return float((v * torch.cos(0.37 * torch.arange(v.numel(), dtype=torch.float64))).sum())


def main():
    torch.set_num_threads(1)          # one thread: the scatter-adds of the reference's CPU backward sum in arrival order across threads (run-to-run differences of 3e-7 relative with 8)
    MG.install_standins()
    from src.model.gaot import GAOT
    from src.model.layers.attn import AttentionConfig, TransformerConfig
    from src.model.layers.magno import MAGNOConfig
    out = {}
    for seed in range(N_SEEDS):
        c = F.draw(seed)
        cfg = NS(args=NS(magno=MAGNOConfig(precompute_edges=True, neighbor_search_method="native", **c.magno),
                         transformer=TransformerConfig(attn_config=AttentionConfig(**c.attn), **c.tf)), latent_tokens_size=c.sizes)
        model = GAOT(input_size=c.cin, output_size=c.cout, config=cfg)
        sd = O.make_state_dict(F.oracle_config(c), c.cin, c.cout, seed=seed)
        res = model.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        b = F.make_batch(c)
        one = lambda cs: {"neighbors_index": cs[0], "neighbors_row_splits": cs[1]}
        vx = c.mode == "vx"
        kw = dict(latent_tokens_coord=b["latent"], xcoord=b["xcoord"], pndata=b["pndata"],
                  encoder_nbrs=[[one(s) for s in row] for row in b["encoder_nbrs"]] if vx else [one(s) for s in b["encoder_nbrs"]],
                  decoder_nbrs=[[one(s) for s in row] for row in b["decoder_nbrs"]] if vx else [one(s) for s in b["decoder_nbrs"]])
        if "query_coord" in b:
            kw["query_coord"] = b["query_coord"]
        if "condition" in b:
            kw["condition"] = b["condition"]
        model.train()
        pred = model(**kw)
        loss = torch.nn.MSELoss()(pred, b["target"])
        model.zero_grad()
        loss.backward()
        out[f"{seed}.pred"] = pred.detach().numpy().astype(np.float32)
        out[f"{seed}.loss"] = np.float64(float(loss))
        names, norms, projs = [], [], []
        for k, prm in model.named_parameters():
            gk = prm.grad if prm.grad is not None else torch.zeros_like(prm)
            names.append(k); norms.append(float(gk.double().norm())); projs.append(proj(gk))
        out[f"{seed}.names"] = np.array(names)
        out[f"{seed}.gnorm"] = np.array(norms)
        out[f"{seed}.gproj"] = np.array(projs)
        print(seed, c.mode, tuple(pred.shape), f"loss {float(loss):.6f}", flush=True)
    np.savez_compressed(os.path.join(HERE, "fuzz_reference_steps.npz"), **out)


if __name__ == "__main__":
    main()
