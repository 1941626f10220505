#!/usr/bin/env python3
"""Generate tests/golden/fuzz_state_dicts.json by building the *imported reference's* GAOT for the first 60 configurations of the randomised
parity sweep (tools/fuzz_parity.py `draw`) and recording its state_dict: parameter names IN ORDER and shapes -- data, not source.
Runs ONLY in the build container (needs /root/reference; same stand-ins for the absent torch_scatter / omegaconf / rotary_embedding_torch as
make_golden.py).  tests/test_fuzz_cpu.py compares gaot_amd's module tree with it.

Usage:  python tests/golden/make_fuzz_state_dicts.py"""
import json
import os
import sys
from types import SimpleNamespace as NS

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import make_golden as MG                                    # noqa: E402  (install_standins only)
from tools import fuzz_parity as F                          # noqa: E402

N_SEEDS = 60


def main():
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
        out[str(seed)] = [[k, list(v.shape)] for k, v in model.state_dict().items()]
    with open(os.path.join(HERE, "fuzz_state_dicts.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", len(out), "state_dict layouts,", sum(len(v) for v in out.values()), "entries")


if __name__ == "__main__":
    main()
