#!/usr/bin/env python3
"""Generate tests/golden/fuzz_reference_rollouts.npz: `autoregressive_predict` of the *imported reference's* GAOT (gaot.py:307-476) on the linear fx
configurations among the first 40 of the randomised parity sweep (tools/fuzz_parity.py: `draw`, `rollout_setup`: 3-6 steps, a random stepper
mode, +- a constant channel, +- conditional norm), weights = oracle.make_state_dict(seed) loaded strictly into the reference model.  Recorded
per seed: the denormalised predictions of every step -- data, not source.  tests/test_fuzz_cpu.py holds the oracle's rollout to it.
Runs ONLY in the build container (needs /root/reference; make_golden.py's stand-ins).

Usage:  python tests/golden/make_fuzz_reference_rollouts.py"""
import os
import sys
from types import SimpleNamespace as NS

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as MG                                    # noqa: E402
from oracle import gaot_oracle as O                         # noqa: E402
from tools import fuzz_parity as F                          # noqa: E402


def seeds():
    return [s for s in range(40) if F.draw(s).mode in ("fx", "fx_own_search") and F.draw(s).magno["transform_type"] == "linear"]


def main():
    torch.set_num_threads(1)          # one thread: the scatter-adds of the reference's CPU backward sum in arrival order across threads (run-to-run differences of 3e-7 relative with 8)
    MG.install_standins()
    from src.model.gaot import GAOT
    from src.model.layers.attn import AttentionConfig, TransformerConfig
    from src.model.layers.magno import MAGNOConfig
    out = {}
    for seed in seeds():
        c = F.draw(seed)
        ro = F.rollout_setup(c)
        cfg = NS(args=NS(magno=MAGNOConfig(precompute_edges=True, neighbor_search_method="native", **c.magno),
                         transformer=TransformerConfig(attn_config=AttentionConfig(**c.attn), **c.tf)), latent_tokens_size=c.sizes)
        model = GAOT(input_size=ro.cin, output_size=ro.udim, config=cfg)
        ocfg = F.oracle_config(c)
        model.load_state_dict(O.make_state_dict(ocfg, ro.cin, ro.udim, seed=seed), strict=True)
        model.eval()
        one = lambda cs: {"neighbors_index": cs[0], "neighbors_row_splits": cs[1]}
        got = model.autoregressive_predict(x_batch=ro.xb, time_indices=ro.ti, t_values=ro.tv, stats=ro.stats, stepper_mode=ro.mode,
                                           latent_tokens_coord=ro.lat, fixed_coord=ro.x, encoder_nbrs=[one(s) for s in ro.enc],
                                           decoder_nbrs=[one(s) for s in ro.dec], use_conditional_norm=ro.cn)
        out[f"{seed}.rollout"] = got.detach().numpy().astype(np.float32)
        print(seed, ro.mode, ro.steps, "cn" if ro.cn else "", "const" if ro.cdim else "", tuple(got.shape), flush=True)
    np.savez_compressed(os.path.join(HERE, "fuzz_reference_rollouts.npz"), **out)


if __name__ == "__main__":
    main()
