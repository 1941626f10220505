"""GAOT = MAGNO encoder -> patch ViT processor -> MAGNO decoder, with the reference's operator API
(reference src/model/gaot.py: constructor, forward / encode / process / decode / autoregressive_predict, and
state_dict keys), running on libgaot_hip.so."""
import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from .. import autograph, ops
from .layers.attn import Transformer
from .layers.magno import MAGNODecoder, MAGNOEncoder


class GAOT(nn.Module):
    def __init__(self, input_size: int, output_size: int, config=None):
        super().__init__()
        magno_cfg = config.args.magno
        tf_cfg = config.args.transformer
        coord_dim = magno_cfg.coord_dim
        if coord_dim not in (2, 3):
            raise ValueError(f"coord_dim must be 2 or 3, got {coord_dim}")
        lts = list(config.latent_tokens_size)
        if len(lts) != coord_dim:
            raise ValueError(f"For {coord_dim}D, latent_tokens_size must have {coord_dim} dimensions, got {len(lts)}")
        self.input_size = input_size
        self.output_size = output_size
        self.coord_dim = coord_dim
        self.node_latent_size = magno_cfg.lifting_channels
        self.patch_size = tf_cfg.patch_size
        self.H, self.W = lts[0], lts[1]
        self.D = lts[2] if coord_dim == 3 else None
        self.encoder = self.init_encoder(input_size, self.node_latent_size, magno_cfg)
        self.processor = self.init_processor(self.node_latent_size, tf_cfg)
        self.decoder = self.init_decoder(output_size, self.node_latent_size, magno_cfg)

    # ---- dtype contract.  The reference trainer casts the model to its configured dtype (base_trainer.py:63-68,173-179:
    # `model.type(dtype)`, dtype float or double).  The HIP kernels compute in fp32 -- the reference's default, and the arithmetic
    # the parity bar is stated in; there is no fp64 (or half) kernel set, and silently computing a "double" model in fp32 would
    # be a wrong answer, not a drop-in.  So the request fails HERE, at the module, with the parameters left untouched in fp32.
    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        bad = next((p.dtype for p in self.parameters() if p.is_floating_point() and p.dtype != torch.float32), None)
        if bad is not None:
            super()._apply(lambda t: t.float() if t.is_floating_point() else t, recurse)
            raise TypeError(f"gaot_amd.GAOT computes in float32 only (requested {bad}): there are no {bad} kernels on the HIP path. "
                            "Keep the trainer's dtype at float32 (the reference default); the parameters were left in float32.")
        return out

    @staticmethod
    def _require_f32(**tensors):
        for name, t in tensors.items():
            if torch.is_tensor(t) and t.is_floating_point() and t.dtype != torch.float32:
                raise TypeError(f"gaot_amd.GAOT.forward: `{name}` is {t.dtype}; the HIP path computes in float32 only "
                                "(cast the inputs, or keep the trainer's dtype at float32)")

    # ---- construction (same order as the reference so a seeded build draws identical weights)
    def init_encoder(self, input_size, latent_size, config):
        return MAGNOEncoder(in_channels=input_size, out_channels=latent_size, config=config)

    def init_processor(self, node_latent_size, config):
        tok = (self.patch_size ** self.coord_dim) * node_latent_size
        self.patch_linear = nn.Linear(tok, tok)
        self.positional_embedding_name = config.positional_embedding
        self.positions = self._get_patch_positions()         # plain attribute, not a buffer (not in state_dict)
        self._pos_emb_cache = None
        return Transformer(input_size=tok, output_size=tok, config=config)

    def init_decoder(self, output_size, latent_size, config):
        return MAGNODecoder(in_channels=latent_size, out_channels=output_size, config=config)

    def _grid_sizes(self) -> List[int]:
        return [self.H, self.W] if self.coord_dim == 2 else [self.H, self.W, self.D]

    def _get_patch_positions(self) -> torch.Tensor:
        axes = [torch.arange(n // self.patch_size, dtype=torch.float32) for n in self._grid_sizes()]
        return torch.stack(torch.meshgrid(*axes, indexing='ij'), dim=-1).reshape(-1, self.coord_dim)

    def _compute_absolute_embeddings(self, positions: torch.Tensor, embed_dim: int) -> torch.Tensor:
        nd = positions.size(1)
        n = embed_dim // (2 * nd)
        inv = 1.0 / (10000 ** (torch.arange(n, dtype=torch.float32, device=positions.device) / n))
        ang = positions[:, :, None] * inv[None, None, :]
        return torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1).view(positions.size(0), -1)

    def _pos_emb(self, device, width: int) -> torch.Tensor:
        """sinusoidal table [S, P^d C]: constant, so built once per device and fused as a row-periodic bias
        into the patch_linear GEMM epilogue (reference recomputes it every call, gaot.py:209-215)."""
        c = self._pos_emb_cache
        if c is None or c.device != device or c.shape[1] != width:
            c = self._compute_absolute_embeddings(self.positions.to(device), width).contiguous()
            if c.shape[1] != width:
                raise ValueError(f"absolute positional embedding width {c.shape[1]} != token width {width} "
                                 f"(needs P^d*C divisible by 2*d)")
            self._pos_emb_cache = c
        return c

    # ---- the three stages
    def encode(self, x_coord, pndata, latent_tokens_coord, encoder_nbrs):
        return self.encoder(x_coord=x_coord, pndata=pndata, latent_tokens_coord=latent_tokens_coord, encoder_nbrs=encoder_nbrs)

    def process(self, rndata: Optional[torch.Tensor] = None, condition: Optional[float] = None, patch_major: bool = False) -> torch.Tensor:
        """`patch_major` (internal, _forward_eager): the rows of `rndata` are the latent points in patch-major order already (and the result
        is expected in it), so the two permutes of gaot.py:182-186 / 222-229 are reshapes"""
        B, n, C = rndata.shape
        P = self.patch_size
        sizes = self._grid_sizes()
        assert n == math.prod(sizes), f"n_regional_nodes ({n}) != {'*'.join(map(str, sizes))}"
        assert all(s % P == 0 for s in sizes), f"latent grid {sizes} must be divisible by P({P})"
        pvol = P ** self.coord_dim
        if patch_major:
            tok = ops.reshaped(rndata.contiguous(), (B, n // pvol, pvol * C))
        else:
            tok = ops.patchify(rndata, sizes, P)                                  # [B, S, P^d C]
        if self.positional_embedding_name == 'absolute':
            tok = ops.linear(tok, self.patch_linear.weight, self.patch_linear.bias, rowbias=self._pos_emb(tok.device, tok.shape[-1]))
            rel = None
        elif self.positional_embedding_name == 'rope':
            tok = ops.linear(tok, self.patch_linear.weight, self.patch_linear.bias)
            rel = self.positions                      # gaot.py:217-218: only its presence matters downstream
        else:
            raise ValueError(f"unknown positional_embedding {self.positional_embedding_name!r}")
        tok = self.processor(tok, condition=condition, relative_positions=rel)
        if patch_major:
            return ops.reshaped(tok.contiguous(), (B, n, C))
        return ops.unpatchify(tok, sizes, P)

    # ---- the latent grid in patch-major order.  patchify / unpatchify move whole C-wide rows (the channel axis stays innermost in both
    # permutes), i.e. they are a fixed permutation of the latent points -- and which latent point carries which number is nobody's business
    # but the caller's neighbour lists'.  So a forward over caller-supplied fx graphs (precompute_edges: the reference's fx trainers build them
    # once, static_trainer.py:120-160) renumbers the latent grid once per graph (plan.renumbered: the encoder's CSR rows re-ordered, the
    # decoder's source indices relabelled, coordinates gathered; all cached) and the four permuting launches of a training step are gone.
    # Every per-row sum keeps its terms and their order, so the forward result is the same bits.
    _PATCH_MAJOR = [os.environ.get("GAOT_PATCH_MAJOR", "1") != "0"]        # A/B switch (tools, tests)

    def _latent_order(self, device):
        """(perm, inv): patch-major row r' holds the caller's latent point perm[r']"""
        c = self.__dict__.get("_latent_order_cache")
        if c is None or c[0].device != device:
            sizes, P, d = self._grid_sizes(), self.patch_size, self.coord_dim
            ids = torch.arange(math.prod(sizes)).view(*[v for s in sizes for v in (s // P, P)])
            perm = ids.permute(*range(0, 2 * d, 2), *range(1, 2 * d, 2)).reshape(-1).to(device)
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(perm.numel(), device=device)
            c = self.__dict__["_latent_order_cache"] = (perm, inv)
        return c

    def _patch_major(self, latent, xcoord, query_coord, encoder_nbrs, decoder_nbrs):
        """(latent coordinates, encoder lists, decoder lists) over the renumbered grid, or None when the forward keeps the caller's numbering:
        vx batches (per-sample lists composed on the device every step), a patch size of 1.  fx graphs handed in by the caller (precompute_edges) and the
        module's own (magno.py:177-180, found over the CALLER's coordinates and cached by shape as ever) are renumbered alike."""
        if not (self._PATCH_MAJOR[0] and self.patch_size > 1 and latent.is_cuda and xcoord.is_cuda and xcoord.dim() == 2
                and (query_coord is None or query_coord.dim() == 2) and latent.dim() == 2
                and latent.shape[0] == math.prod(self._grid_sizes())):
            return None
        enc, dec = self.encoder, self.decoder
        if bool(enc.precompute_edges) != bool(dec.precompute_edges):
            return None
        if self.training and (enc.sampling_strategy is not None or dec.sampling_strategy is not None):
            return None              # neighbour sub-sampling draws per edge POSITION: the draws (and plan.DROP_RECORD) stay in the caller's numbering
        if not enc.precompute_edges:
            if latent.shape[1] != self.coord_dim or xcoord.shape[1] != self.coord_dim:
                return None          # (the stages raise the reference's errors)
            encoder_nbrs = enc._compute_neighbors(xcoord, latent, 'fx')
            decoder_nbrs = dec._compute_neighbors(latent, xcoord if query_coord is None else query_coord, 'fx')
        ns = len(enc.scales)
        for nb in (encoder_nbrs, decoder_nbrs):
            if not (isinstance(nb, (list, tuple)) and len(nb) == ns
                    and all(isinstance(d, dict) and torch.is_tensor(d.get("neighbors_index")) and d["neighbors_index"].is_cuda
                            and torch.is_tensor(d.get("neighbors_row_splits")) and d["neighbors_row_splits"].is_cuda for d in nb)):
                return None
        if any(int(d["neighbors_row_splits"].numel()) != latent.shape[0] + 1 for d in encoder_nbrs):
            return None              # (not this latent grid's list: the encoder raises as ever)
        from ..plan import renumbered
        from .layers.magno import RenumberedLists
        perm, inv = self._latent_order(latent.device)
        if getattr(self, "_auto_graph_bypass", False):
            # autograph's forward (warm-up, capture): `latent` is its static coordinate buffer, whose BYTES follow the caller's at the same object
            # and version (autograph._sync_coordinates raises the plans' refresh flag when they change).  The renumbered coordinates live in a
            # static buffer of their own, re-gathered by a launch that is part of the captured forward (ahead of the flag-guarded refresh
            # kernels that read it); written through .data, so the plans' identity-keyed arrays see one object at one version
            store = self.__dict__.setdefault("_latent_pm_static", {})
            ent = store.get(id(latent))
            if ent is None or ent[0] is not latent:
                if len(store) > 8:
                    store.clear()
                ent = store[id(latent)] = (latent, torch.empty_like(latent, memory_format=torch.contiguous_format))
            torch.index_select(latent.detach(), 0, perm, out=ent[1].data)
            lat_pm = ent[1]
            if torch.cuda.is_current_stream_capturing():
                self.__dict__.setdefault("_graph_keep", []).append(ent)
        else:
            key = (latent._version, id(perm))
            hit = self.__dict__.get("_latent_pm")
            if hit is None or hit[0] is not latent or hit[1] != key:
                hit = self.__dict__["_latent_pm"] = (latent, key, latent.detach()[perm].contiguous(), perm)
            if torch.cuda.is_current_stream_capturing():          # read by address from the graph being captured: keep it past the cache entry
                self.__dict__.setdefault("_graph_keep", []).append(hit)
            lat_pm = hit[2]
        return (lat_pm, RenumberedLists(renumbered(d, "queries", perm, inv) for d in encoder_nbrs),
                RenumberedLists(renumbered(d, "sources", perm, inv) for d in decoder_nbrs))

    def decode(self, latent_tokens_coord, rndata, query_coord, decoder_nbrs):
        return self.decoder(latent_tokens_coord=latent_tokens_coord, rndata=rndata, query_coord=query_coord, decoder_nbrs=decoder_nbrs)

    def forward(self, latent_tokens_coord: torch.Tensor, xcoord: torch.Tensor, pndata: torch.Tensor,
                query_coord: Optional[torch.Tensor] = None, encoder_nbrs: Optional[list] = None,
                decoder_nbrs: Optional[list] = None, condition: Optional[float] = None) -> torch.Tensor:
        self._require_f32(latent_tokens_coord=latent_tokens_coord, xcoord=xcoord, pndata=pndata, query_coord=query_coord,
                          condition=condition)
        # an unchanged eager training loop (the reference trainer's) on fixed shapes: forward and backward as hipGraph replays
        if autograph.eligible(self, latent_tokens_coord, xcoord, pndata, query_coord, encoder_nbrs, decoder_nbrs, condition):
            out = autograph.run(self, latent_tokens_coord, xcoord, pndata, condition, encoder_nbrs, decoder_nbrs)
            if out is not None:
                return out
        return self._forward_eager(latent_tokens_coord, xcoord, pndata, query_coord, encoder_nbrs, decoder_nbrs, condition)

    def _forward_eager(self, latent_tokens_coord, xcoord, pndata, query_coord=None, encoder_nbrs=None, decoder_nbrs=None,
                       condition=None) -> torch.Tensor:
        if pndata.is_cuda and ops.wants_amax():
            # magnitude words of the fp16-piece products: a fresh arena for this pass, every weight's word (and planes) in two launches.
            # Inference over unchanged weights (validation loops, the steps of an autoregressive rollout) keeps the previous pass's table.
            if getattr(self, "_amax_lists", None) is None:
                self._amax_lists = (list(self.parameters()),
                                    [g for m in self.modules() if hasattr(m, "fused_weight_groups") for g in m.fused_weight_groups()])
            if not torch.is_grad_enabled() and ops.weights_current(self._amax_lists[0]):
                ops.begin_pass(keep_weights=True)
                if torch.cuda.is_current_stream_capturing():
                    ops.pin_weight_table()
            else:
                ops.begin_pass()
                ops.refresh_weight_amax(*self._amax_lists)
        pm = self._patch_major(latent_tokens_coord, xcoord, query_coord, encoder_nbrs, decoder_nbrs)
        if pm is not None:
            latent_tokens_coord, encoder_nbrs, decoder_nbrs = pm
        rn = self.encode(x_coord=xcoord, pndata=pndata, latent_tokens_coord=latent_tokens_coord, encoder_nbrs=encoder_nbrs)
        rn = ops.cut(rn)                      # staged backward (data-parallel training): encoder gradients complete last
        rn = self.process(rndata=rn, condition=condition, patch_major=pm is not None)
        if query_coord is None:
            query_coord = xcoord
        return self.decode(latent_tokens_coord=latent_tokens_coord, rndata=rn, query_coord=query_coord, decoder_nbrs=decoder_nbrs)

    def backward_phases(self) -> List[List[nn.Parameter]]:
        """parameters in backward-COMPLETION order, one list per stretch between cut points (forward(): after the encoder;
        Transformer.forward(): after every block but the last): [last block + output_proj + decoder], ..., [first block +
        input_proj + patch_linear], [encoder].  trainer.TrainStep reduces each list's gradients while the next one is computed."""
        blocks = self.processor.blocks_in_order()
        phases = [list(b.parameters()) for b in reversed(blocks)]
        if isinstance(self.processor.output_proj, nn.Linear):
            phases[0] += list(self.processor.output_proj.parameters())
        phases[0] += list(self.decoder.parameters())
        if isinstance(self.processor.input_proj, nn.Linear):
            phases[-1] += list(self.processor.input_proj.parameters())
        phases[-1] += list(self.patch_linear.parameters())
        phases.append(list(self.encoder.parameters()))
        return phases

    # ---- rollout (reference gaot.py:307-476)
    def autoregressive_predict(self, x_batch: torch.Tensor, time_indices: np.ndarray, t_values: np.ndarray, stats: Dict,
                               stepper_mode: str = "output", latent_tokens_coord: Optional[torch.Tensor] = None,
                               fixed_coord: Optional[torch.Tensor] = None, encoder_nbrs: Optional[List] = None,
                               decoder_nbrs: Optional[List] = None, use_conditional_norm: bool = False) -> torch.Tensor:
        if stepper_mode not in ("output", "residual", "time_der"):
            raise ValueError(f"Unsupported stepper_mode: {stepper_mode}")
        dev, dt_ = x_batch.device, x_batch.dtype
        B, N, _ = x_batch.shape
        u_mean, u_std = stats["u"]["mean"].to(dev), stats["u"]["std"].to(dev)
        udim = u_mean.shape[0]
        cdim = stats["c"]["mean"].shape[0] if "c" in stats else 0
        static = x_batch[..., udim:udim + cdim] if cdim > 0 else None
        state = x_batch[..., :udim]
        graphs = dict(encoder_nbrs=encoder_nbrs, decoder_nbrs=decoder_nbrs) \
            if (encoder_nbrs is not None and decoder_nbrs is not None) else {}
        aux = {k: (stats[k]["mean"].to(dev), stats[k]["std"].to(dev)) for k in ("res", "der") if k in stats}
        runner = _RolloutRunner(self, latent_tokens_coord, fixed_coord, graphs, use_conditional_norm) if x_batch.is_cuda else None
        n_steps = len(time_indices) - 1
        if runner is not None and dt_ == torch.float32 and n_steps > 0:
            return self._rollout_device(runner, state, static, u_mean, u_std, aux, stats, time_indices, t_values, stepper_mode,
                                        use_conditional_norm)
        preds = []
        with torch.no_grad():
            for i in range(1, len(time_indices)):
                t0 = t_values[time_indices[i - 1]]
                dt = t_values[time_indices[i]] - t0
                t0n = (t0 - stats["start_time"]["mean"]) / stats["start_time"]["std"]
                dtn = (dt - stats["time_diffs"]["mean"]) / stats["time_diffs"]["std"]
                cols = [state] + ([static] if static is not None else [])
                cols += [torch.full((B, N, 1), float(t0n), dtype=dt_, device=dev),
                         torch.full((B, N, 1), float(dtn), dtype=dt_, device=dev)]
                xin = torch.cat(cols, dim=-1)
                if use_conditional_norm:          # the dt column is dropped, the time enters through `condition`
                    pn, cond = xin[..., :-1].contiguous(), xin[..., 0, -2:-1].contiguous()
                else:
                    pn, cond = xin, None
                pred = self.forward(latent_tokens_coord=latent_tokens_coord, xcoord=fixed_coord, pndata=pn, condition=cond, **graphs)
                if stepper_mode == "output":
                    den = pred * u_std + u_mean
                elif stepper_mode == "residual":
                    den = (state * u_std + u_mean) + (pred * aux["res"][1] + aux["res"][0])
                else:
                    den = (state * u_std + u_mean) + float(dt) * (pred * aux["der"][1] + aux["der"][0])
                preds.append(den)
                state = (den - u_mean) / u_std
        return torch.stack(preds, dim=1)


    def _rollout_device(self, runner, state, static, u_mean, u_std, aux, stats, time_indices, t_values, stepper_mode, cond_norm):
        """the rollout loop on the device: per step ONE launch assembles the input rows straight into the
        captured forward's static buffer, the forward is a hipGraph replay, ONE launch applies the stepper mode, de-normalises into
        the output slab and re-normalises the running state (gaot.py:371-388, 432, 436-476)."""
        import ctypes as C
        from .. import _lib as L
        lib = L.load()
        B, N, U = state.shape
        S = 0 if static is None else static.shape[-1]
        dev = state.device
        state = state.contiguous().clone()
        stat = None if static is None else static.contiguous()
        n_steps = len(time_indices) - 1
        # step-major slab: every step's [B, N, U] block is contiguous, so the stepper kernel writes it in place (no per-step strided
        # copy); ONE permuting copy at the end gives the reference's torch.stack(preds, dim=1) layout
        out = torch.empty(n_steps, B, N, U, device=dev, dtype=torch.float32)
        pn = torch.empty(B, N, U + S + (1 if cond_norm else 2), device=dev, dtype=torch.float32)
        mode = {"output": 0, "residual": 1, "time_der": 2}[stepper_mode]
        if self.output_size != U:
            raise ValueError(f"autoregressive_predict: the model predicts {self.output_size} channels but stats['u'] describes {U}")

        def per_channel(t, what):
            """[U] fp32 statistics for the fused stepper kernel (it indexes [c], c < U): scalars / [1] broadcast like the eager
            expression `pred * std + mean` would, anything else is a shape error here rather than an out-of-bounds read there"""
            t = t.to(device=dev, dtype=torch.float32).reshape(-1)
            if t.numel() == 1:
                t = t.expand(U)
            if t.numel() != U:
                raise ValueError(f"autoregressive_predict: stats {what} has {t.numel()} entries, expected {U} (or 1)")
            return t.contiguous()

        a_mean = a_std = None
        if mode:
            k = "res" if mode == 1 else "der"
            if k not in aux:
                raise KeyError(k)          # the reference indexes stats['res'] / stats['der'] directly (gaot.py:448-470)
            a_mean, a_std = per_channel(aux[k][0], f"['{k}']['mean']"), per_channel(aux[k][1], f"['{k}']['std']")
        um, us = per_channel(u_mean, "['u']['mean']"), per_channel(u_std, "['u']['std']")
        with torch.no_grad():
            for i in range(1, len(time_indices)):
                t0 = t_values[time_indices[i - 1]]
                dt = t_values[time_indices[i]] - t0
                t0n = float((t0 - stats["start_time"]["mean"]) / stats["start_time"]["std"])
                dtn = float((dt - stats["time_diffs"]["mean"]) / stats["time_diffs"]["std"])
                # straight into the captured forward's static input once it exists (the first step captures from `pn`)
                xin = runner.static_input(pn.shape, (B, 1) if cond_norm else None)
                xin = pn if xin is None else xin
                L.check(lib.gaot_rollout_input(ops._p(state), U, ops._p(stat), S, t0n, dtn, 1 if cond_norm else 2, B * N, ops._p(xin),
                                               ops._stream()), "gaot_rollout_input")
                cond = torch.full((B, 1), t0n, dtype=torch.float32, device=dev) if cond_norm else None
                pred = runner(xin, cond, clone=False)
                if pred.shape != (B, N, U):
                    raise ValueError(f"autoregressive_predict: forward returned {tuple(pred.shape)}, expected {(B, N, U)}")
                L.check(lib.gaot_rollout_update(ops._p(pred.contiguous()), ops._p(state), U, ops._p(um), ops._p(us), ops._p(a_mean), ops._p(a_std),
                                                float(dt), mode, B * N, ops._p(out[i - 1]), ops._stream()), "gaot_rollout_update")
        return out.permute(1, 0, 2, 3).contiguous()


class _RolloutRunner:
    """One forward per rollout step as a hipGraph replay: geometry, neighbour lists, kernel values,
    attention weights and the geoembed row-bias are step-invariant (cached under no_grad); only `pndata` / `condition`
    change, so the step is captured once per (shape, weights version) and replayed with new inputs."""

    def __init__(self, model, latent, coord, graphs, cond):
        self.model, self.latent, self.coord, self.graphs, self.cond = model, latent, coord, graphs, cond
        self.key = None
        self.graph = None
        self._ver = None

    def _versions(self):
        # once per runner = once per autoregressive_predict call (a no_grad loop: the weights cannot change between its steps); the walk
        # over the parameters is host time the device waits for between a step's launches
        if self._ver is None:
            self._ver = tuple(p._version for p in self.model.parameters()) + (ops.weights_generation(),)
        return self._ver

    def _key(self, pn_shape, cond_shape):
        return (tuple(pn_shape), None if cond_shape is None else tuple(cond_shape), id(self.latent), id(self.coord), self._versions())

    def static_input(self, pn_shape, cond_shape=None) -> Optional[torch.Tensor]:
        """the captured forward's own input buffer for these shapes (None before the first capture): a caller that assembles its rows
        there hands the same tensor to __call__ and saves the copy"""
        cache = getattr(self.model, "_rollout_graph", None)
        if cache is None or cache["key"] != self._key(pn_shape, cond_shape):
            return None
        return cache["x"]

    def __call__(self, pn: torch.Tensor, cond: Optional[torch.Tensor], clone: bool = True):
        m = self.model
        key = self._key(pn.shape, None if cond is None else cond.shape)
        cache = getattr(m, "_rollout_graph", None)
        if cache is None or cache["key"] != key:
            x = pn.clone()
            c = None if cond is None else cond.clone()
            kw = dict(latent_tokens_coord=self.latent, xcoord=self.coord, pndata=x, condition=c, **self.graphs)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                m.forward(**kw)              # warm-up: builds plans / inference caches outside the capture
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                y = m.forward(**kw)
            cache = {"key": key, "graph": g, "x": x, "c": c, "y": y, "keep": (self.latent, self.coord)}
            m._rollout_graph = cache
        if cache["x"].data_ptr() != pn.data_ptr():          # (the rollout loop writes its rows into the static buffer itself)
            cache["x"].copy_(pn)
        if cond is not None:
            cache["c"].copy_(cond)
        cache["graph"].replay()
        return cache["y"].clone() if clone else cache["y"]
