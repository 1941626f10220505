"""not gpu: the host side of the randomised parity sweep (tools/fuzz_parity.py).  For 60 random configurations of the operator API the
module tree built by gaot_amd has exactly the reference's parameters -- names, order within the state_dict and shapes as
oracle.make_state_dict lays them out from gaot.py:21-90, magno.py:87-156 / 423-492, attn.py:239-288 -- so a checkpoint of the reference loads
whatever options it was trained with; and the oracle (the checker's half of the sweep) runs a train step on the configurations
tests/test_fuzz_gpu.py holds the HIP path to."""
import os
from types import SimpleNamespace as NS

import pytest

from oracle import gaot_oracle as O
from tools import fuzz_parity as F


def _pair(c):
    from gaot_amd.model.gaot import GAOT
    from gaot_amd.model.layers.attn import AttentionConfig, TransformerConfig
    from gaot_amd.model.layers.magno import MAGNOConfig
    m, t, a = c.magno, c.tf, c.attn
    model = GAOT(c.cin, c.cout, NS(args=NS(magno=MAGNOConfig(precompute_edges=True, **m),
                                           transformer=TransformerConfig(attn_config=AttentionConfig(**a), **t)), latent_tokens_size=c.sizes))
    ocfg = O.OracleConfig(coord_dim=c.d, radius=m["radius"], hidden_size=m["hidden_size"], mlp_layers=m["mlp_layers"],
                          lifting_channels=m["lifting_channels"], scales=m["scales"], use_scale_weights=m["use_scale_weights"],
                          use_attention=m["use_attention"], attention_type=m["attention_type"], use_geoembed=m["use_geoembed"],
                          embedding_method=m["embedding_method"], pooling=m["pooling"], transform_type=m["transform_type"],
                          node_embedding=m["node_embedding"], precompute_edges=True, patch_size=t["patch_size"], tf_hidden_size=t["hidden_size"],
                          use_attn_norm=t["use_attn_norm"], use_ffn_norm=t["use_ffn_norm"], num_layers=t["num_layers"],
                          positional_embedding=t["positional_embedding"], use_long_range_skip=t["use_long_range_skip"],
                          ffn_multiplier=t["ffn_multiplier"], num_heads=a["num_heads"], num_kv_heads=a["num_kv_heads"],
                          use_conditional_norm=a["use_conditional_norm"], latent_tokens_size=c.sizes)
    return model, ocfg


def test_random_configurations_have_the_imported_references_state_dict():
    """names IN ORDER and shapes against tests/golden/fuzz_state_dicts.json = the state_dict of the imported reference's own GAOT for the same 60
    configurations (tests/golden/make_fuzz_state_dicts.py wrote it in the build container)"""
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_state_dicts.json")) as f:
        ref_all = json.load(f)
    assert len(ref_all) == 60
    for seed in range(60):
        model, _ = _pair(F.draw(seed))
        mine = [[k, list(v.shape)] for k, v in model.state_dict().items()]
        ref = ref_all[str(seed)]
        assert [k for k, _ in mine] == [k for k, _ in ref], (seed, [a for a, b in zip(mine, ref) if a[0] != b[0]][:4])
        assert mine == ref, (seed, [(a, b) for a, b in zip(mine, ref) if a != b][:4])


def test_random_configurations_have_the_reference_parameters():
    for seed in range(60):
        c = F.draw(seed)
        model, ocfg = _pair(c)
        mine = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        ref = {k: tuple(v.shape) for k, v in O.make_state_dict(ocfg, c.cin, c.cout).items()}
        assert mine == ref, (seed, sorted(set(mine) ^ set(ref))[:6], [(k, mine[k], ref[k]) for k in mine if k in ref and mine[k] != ref[k]][:4])


@pytest.mark.parametrize("seed", [42, 49, 58])
def test_the_oracle_runs_the_sweeps_regression_seeds(seed, monkeypatch):
    import torch
    monkeypatch.setenv("FUZZ_ORACLE_ONLY", "1")
    ok, info = F.run(F.draw(seed), torch.device("cpu"))
    assert ok and info["loss_ref"] == info["loss_ref"] and 0.0 < info["loss_ref"] < 100.0, info


def _proj(g):
    import torch
    v = g.detach().double().reshape(-1)
    return float((v * torch.cos(0.37 * torch.arange(v.numel(), dtype=torch.float64))).sum())


@pytest.mark.parametrize("seed", range(24))
def test_the_oracle_reproduces_the_imported_references_step_on_random_configurations(seed):
    """tests/golden/fuzz_reference_steps.npz = one trainer step of the imported reference's GAOT per configuration (make_fuzz_reference_steps.py):
    the oracle, with the same seeded weights and the same batch, must give its prediction, loss and every parameter gradient (norm and a fixed
    cosine projection).  Pins the sweep's checker to the reference across the sweep's option space."""
    import numpy as np
    import torch
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_reference_steps.npz"), allow_pickle=False)
    c = F.draw(seed)
    ocfg = F.oracle_config(c)
    sd = O.make_state_dict(ocfg, c.cin, c.cout, seed=seed)
    loss, grads, _, _, pred = O.train_step(sd, ocfg, F.make_batch(c), return_pred=True)
    ref_pred = torch.from_numpy(z[f"{seed}.pred"])
    assert float((pred - ref_pred).norm() / ref_pred.norm()) < 5e-6
    assert abs(float(loss) - float(z[f"{seed}.loss"])) < 5e-6 * abs(float(z[f"{seed}.loss"]))
    names, gnorm, gproj = [str(k) for k in z[f"{seed}.names"]], z[f"{seed}.gnorm"], z[f"{seed}.gproj"]
    assert set(names) == set(grads)
    top = float(gnorm.max())
    for k, n_ref, p_ref in zip(names, gnorm, gproj):
        scale = max(float(n_ref), 1e-4 * top)
        assert abs(float(grads[k].double().norm()) - float(n_ref)) < 2e-4 * scale, (k, float(grads[k].norm()), float(n_ref))
        assert abs(_proj(grads[k]) - float(p_ref)) < 2e-4 * scale, (k, _proj(grads[k]), float(p_ref))


def _rollout_seeds():
    return [s for s in range(40) if F.draw(s).mode in ("fx", "fx_own_search") and F.draw(s).magno["transform_type"] == "linear"]


@pytest.mark.parametrize("seed", _rollout_seeds())
def test_the_oracle_reproduces_the_imported_references_rollout_on_random_configurations(seed):
    """tests/golden/fuzz_reference_rollouts.npz = autoregressive_predict of the imported reference's GAOT (make_fuzz_reference_rollouts.py): the
    oracle's rollout -- the checker of the GPU sweep's rollouts -- gives every step of it (3-6 steps, all three stepper modes, +- a constant
    channel, +- conditional norm)"""
    import numpy as np
    import torch
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_reference_rollouts.npz"), allow_pickle=False)
    c = F.draw(seed)
    ro = F.rollout_setup(c)
    ocfg = F.oracle_config(c)
    sd = O.make_state_dict(ocfg, ro.cin, ro.udim, seed=seed)
    got = O.autoregressive_predict(sd, ocfg, ro.xb, ro.ti, ro.tv, ro.stats, ro.mode, ro.lat, ro.x, use_conditional_norm=ro.cn,
                                   encoder_nbrs=ro.enc, decoder_nbrs=ro.dec)
    ref = torch.from_numpy(z[f"{seed}.rollout"])
    assert got.shape == ref.shape == (c.B, ro.steps, c.N, ro.udim)
    for i in range(ro.steps):
        assert float((got[:, i] - ref[:, i]).norm() / ref[:, i].norm()) < 1e-5, (seed, i)
