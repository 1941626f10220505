"""Loader for tests/golden/*.npz (written by tests/golden/make_golden.py from the imported reference)."""
import ast
import os

import numpy as np
import torch

from oracle.gaot_oracle import OracleConfig

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

MODEL_CASES = ["fx2d_base", "fx2d_base_s1", "fx2d_zero_deg", "fx2d_headdim32", "fx2d_inproj", "vx2d",
               "ms_mean", "ms_weighted", "attn_dot", "no_attn_mean", "no_geoembed", "nonlinear",
               "node_embed", "pointnet", "fx3d", "even_layers", "linear_kernelonly", "nonlinear_kernelonly", "rope", "pointnet_mean"]


class Golden:
    def __init__(self, case):
        self.case = case
        z = np.load(os.path.join(GOLDEN_DIR, f"{case}.npz"), allow_pickle=False)
        self.raw = {k: z[k] for k in z.files}
        self.magno = ast.literal_eval(str(self.raw["meta.magno"]))
        self.transformer = ast.literal_eval(str(self.raw["meta.transformer"]))
        self.attn = ast.literal_eval(str(self.raw["meta.attn"]))

    def t(self, key):
        return torch.from_numpy(np.array(self.raw[key]))

    def has(self, key):
        return key in self.raw

    def group(self, prefix):
        return {k[len(prefix):]: torch.from_numpy(np.array(v)) for k, v in self.raw.items() if k.startswith(prefix)}

    @property
    def state_dict(self):
        return self.group("w.")

    @property
    def latent_tokens_size(self):
        lat = self.raw["in.latent"]
        d = lat.shape[1]
        n = round(lat.shape[0] ** (1.0 / d))
        return [n] * d

    def oracle_config(self) -> OracleConfig:
        m, t, a = self.magno, self.transformer, self.attn
        return OracleConfig(
            coord_dim=m.get("coord_dim", 2), radius=m["radius"], hidden_size=m["hidden_size"],
            mlp_layers=m["mlp_layers"], lifting_channels=m["lifting_channels"], scales=m.get("scales", [1.0]),
            use_scale_weights=m.get("use_scale_weights", False), use_attention=m.get("use_attention", True),
            attention_type=m.get("attention_type", "cosine"), use_geoembed=m.get("use_geoembed", True),
            embedding_method=m.get("embedding_method", "statistical"), pooling=m.get("pooling", "max"),
            transform_type=m.get("transform_type", "linear"), node_embedding=m.get("node_embedding", False),
            precompute_edges=m.get("precompute_edges", False),
            patch_size=t["patch_size"], tf_hidden_size=t["hidden_size"], num_layers=t.get("num_layers", 3),
            num_heads=a["num_heads"], num_kv_heads=a["num_kv_heads"],
            use_conditional_norm=a.get("use_conditional_norm", False),
            positional_embedding=t.get("positional_embedding", "absolute"),
            latent_tokens_size=self.latent_tokens_size)

    def csr_lists(self):
        """(encoder_nbrs, decoder_nbrs) in the reference's list layout, or (None, None) when absent."""
        nsc = len(self.magno.get("scales", [1.0]))
        if self.has("csr.enc.b0.s0.index"):
            B = self.raw["in.pndata"].shape[0]
            mk = lambda side: [[(self.t(f"csr.{side}.b{b}.s{s}.index"), self.t(f"csr.{side}.b{b}.s{s}.splits"))
                                for s in range(nsc)] for b in range(B)]
            return mk("enc"), mk("dec")
        if self.has("csr.enc.s0.index"):
            mk = lambda side: [(self.t(f"csr.{side}.s{s}.index"), self.t(f"csr.{side}.s{s}.splits")) for s in range(nsc)]
            return mk("enc"), mk("dec")
        return None, None


def rel_l2(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    den = b.norm().item()
    return (a - b).norm().item() / (den if den > 0 else 1.0)


STATS_GATED = ("encoder.geoembed.mlp.0.weight", "encoder.geoembed.mlp.0.bias", "decoder.geoembed.mlp.0.weight", "decoder.geoembed.mlp.0.bias")


class StatsGates:
    """tests/golden/c2_stats_gates.npz: the REFERENCE at the bench configuration in float32 and in float64 on identical weights
    (make_golden.run_c2_stats_gates): losses, the relative movement of the prediction and of every gradient tensor between the
    two, and the full gradients of the four tensors behind the geometry statistics' ReLU gates in both precisions."""

    def __init__(self):
        z = np.load(os.path.join(GOLDEN_DIR, "c2_stats_gates.npz"), allow_pickle=False)
        self.raw = {k: z[k] for k in z.files}
        self.move = {k[5:]: float(v) for k, v in self.raw.items() if k.startswith("move.")}
        self.top = float(self.raw["grad_norm_top"])
        self.g32 = {k: torch.from_numpy(np.array(self.raw[f"g32.{k}"])) for k in STATS_GATED}
        self.g64 = {k: torch.from_numpy(np.array(self.raw[f"g64.{k}"])) for k in STATS_GATED}

    def same_weights(self, sd) -> bool:
        """the seeded build drew the weights the fixture was made with (sum and norm of every tensor, in float64)"""
        for k, v in sd.items():
            want = self.raw[f"wsum.{k}"]
            got = np.array([float(v.double().sum()), float(v.double().norm())])
            if not np.allclose(got, want, rtol=1e-12, atol=1e-12):
                return False
        return True

    def err(self, grads, ref) -> dict:
        """per gated tensor: rel-L2 against `ref` (self.g32 or self.g64), denominator floored as in the parity tests"""
        return {k: float((grads[k].detach().cpu().double() - ref[k].double()).norm()) / max(float(ref[k].double().norm()), 1e-3 * self.top)
                for k in STATS_GATED}


# ---------------------------------------------------------------------------------------------------------------------------------
# Gradient parity WITHOUT a floor on the denominator.  Rounds 1-5 divided a tensor's error by max(its norm, 1e-3 x the model's largest gradient
# norm): a small tensor could be lost altogether and pass.  What a small tensor can legitimately be compared against is the rounding the
# reference's OWN fp32 arithmetic leaves on it: the oracle evaluated in float32 against the oracle evaluated in float64 end to end, same
# weights, same batch (`fp32_noise`).  A tensor passes when
#     ||got - ref||  <=  max( tol * ||ref|| ,  3 * ||ref32 - ref64|| + 1e-9 * max_k ||ref_k|| )
# i.e. within `tol` relative, or -- for tensors at or below their own fp32 rounding (a key bias under a softmax: zero in exact arithmetic) --
# within three times the distance the fp32 reference itself keeps from float64 (the absolute term only covers tensors BOTH evaluations hit
# exactly: 1e-9 of the model's gradient scale, six orders below the old floor).
# ---------------------------------------------------------------------------------------------------------------------------------
def fp32_noise(sd, cfg, batch, grads32=None, **step_kw):
    """k -> ||g32_k - g64_k||: the reference algorithm's own fp32 rounding per gradient tensor (one float64 oracle pass on the CPU)"""
    from oracle import gaot_oracle as O
    dbl = lambda v: v.double() if torch.is_tensor(v) and v.is_floating_point() else v
    if grads32 is None:
        grads32 = O.train_step(sd, cfg, batch, **step_kw)[1]
    grads64 = O.train_step({k: dbl(v) for k, v in sd.items()}, cfg, {k: dbl(v) for k, v in batch.items()}, **step_kw)[1]
    return {k: float((grads32[k].double() - grads64[k]).norm()) for k in grads64}


def unfloored_ratio(got: dict, ref: dict, noise: dict, tol: float) -> dict:
    """k -> ||got_k - ref_k|| / bar_k with the bar above: <= 1 passes.  `got` may lack tensors (no gradient): compared as zeros."""
    top = max(float(v.double().norm()) for v in ref.values())
    out = {}
    for k, r in ref.items():
        r = r.double()
        g = got[k].detach().cpu().double() if got.get(k) is not None else torch.zeros_like(r)
        bar = max(tol * float(r.norm()), 3.0 * noise[k] + 1e-9 * top)
        out[k] = float((g - r).norm()) / bar
    return out
