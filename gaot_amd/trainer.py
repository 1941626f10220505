"""Training-step harness with the semantics of the reference trainer step
(static_trainer.py:160-178, optimizers.py:247-257, loss base_trainer.py:71):

    zero_grad -> pred = model(...) -> MSELoss(mean) -> backward -> AdamW.step

plus what the reference lacks: data-parallel training.  One process per GPU; all parameter
gradients live in ONE flat fp32 buffer (param.grad are views), so the exchange is a single RCCL all-reduce
(13.6 MB at the example config) per step; parameters are broadcast from rank 0 at start (the reference seeds
with seed+rank and never synchronises).  With static shapes the forward+backward and the optimizer update are
captured as hipGraphs (geometry-only work is cached in GeometryPlans and stays outside the graph).

With more than one rank the backward pass is STAGED: the model marks cut points (`ops.cut`, after the encoder
and between transformer blocks) and lists its parameters in backward-completion order (`backward_phases()`: decoder +
last block first, encoder last).  The flat gradient buffer is laid out in that order, one contiguous slice per phase;
after the backward of phase k has been enqueued its slice is all-reduced asynchronously (RCCL runs it on its own stream)
while the main stream already runs the backward of phase k+1 -- one captured hipGraph per phase, nothing waits on the host.
"""
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


def shard_indices(n_samples: int, rank: int, world: int, epoch: int = 0, shuffle: bool = True, seed: int = 0) -> List[int]:
    """DistributedSampler-equivalent: same permutation on every rank, padded to a multiple of `world`, strided."""
    g = torch.Generator().manual_seed(seed + epoch)
    order = torch.randperm(n_samples, generator=g).tolist() if shuffle else list(range(n_samples))
    total = ((n_samples + world - 1) // world) * world
    if n_samples > 0 and total > n_samples:      # repeat the list as often as needed (world > n_samples): no rank stays empty
        order = (order * (-(-total // n_samples)))[:total]
    return order[rank:total:world]


class FlatGradBucket:
    """All gradients in one contiguous fp32 buffer (RCCL all-reduce payload, one fused optimizer pass).

    Autograd is left to hand over freshly produced gradient tensors (param.grad is None before backward, so
    AccumulateGrad steals instead of launching one add per parameter); `pack()` then gathers them with ONE
    multi-tensor copy and re-points param.grad at views of the flat buffer for the all-reduce / optimizer.

    `phases` (optional): lists of parameters in backward-COMPLETION order.  The buffer is laid out phase by phase
    (`segments[k]` = the contiguous slice of phase k), so a phase's gradients can be reduced while later phases of the
    backward pass are still running."""

    ALIGN = 64      # floats: every parameter (or fused group) starts on a 256-byte boundary -> 16-byte vector kernels apply

    def __init__(self, params: List[torch.nn.Parameter], groups: Optional[List[List[torch.nn.Parameter]]] = None,
                 phases: Optional[List[List[torch.nn.Parameter]]] = None):
        """`groups`: lists of parameters that must sit back to back (in the given order, no padding) so that a fused
        GEMM can read them as ONE matrix (q|k|v, w1|w3: ops.adjacent_rows) instead of concatenating every step."""
        params = [p for p in params if p.requires_grad]
        self.model_order = list(params)           # the order torch optimizers index parameters by (checkpoint compatibility)
        pos = {id(p): i for i, p in enumerate(params)}
        phase_of = {}
        for k, ph in enumerate(phases or []):
            for p in ph:
                if id(p) in pos:
                    phase_of[id(p)] = k
        n_ph = (max(phase_of.values()) + 1) if phase_of else 1
        for p in params:                          # parameters the model did not list complete with the LAST phase
            phase_of.setdefault(id(p), n_ph - 1)
        follow, skip = {}, set()
        for g in groups or []:
            if (len(g) > 1 and all(id(p) in pos for p in g) and not any(id(p) in skip or id(p) in follow for p in g)
                    and len({phase_of[id(p)] for p in g}) == 1):
                follow[id(g[0])] = list(g[1:])
                skip.update(id(p) for p in g[1:])
        self.params, self.offsets, self.phase, self.segments = [], [], [], []
        off = 0
        for k in range(n_ph):
            off = -(-off // self.ALIGN) * self.ALIGN
            start = off
            for p in params:
                if id(p) in skip or phase_of[id(p)] != k:
                    continue
                off = -(-off // self.ALIGN) * self.ALIGN
                for q in [p] + follow.get(id(p), []):
                    self.params.append(q)
                    self.offsets.append(off)
                    self.phase.append(k)
                    off += q.numel()
            self.segments.append((start, off))
        self.numel = -(-off // 4) * 4
        self.segments[-1] = (self.segments[-1][0], self.numel)
        ref = self.params[0]
        self.flat = torch.zeros(self.numel, dtype=ref.dtype, device=ref.device)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self.offsets)]
        self._dirty = [False] * len(self.params)      # view holds a gradient from an earlier step

    @property
    def n_phases(self) -> int:
        return len(self.segments)

    def clear(self):
        for p in self.params:
            p.grad = None
        if self.flat.is_cuda:         # weight-gradient GEMMs of the HIP ops write straight into their views (ops._claim)
            from . import ops
            if ops.grad_slot_of(self.params[0]) is not self.views[0]:
                ops.register_grad_slots(self.params, self.views)
            ops.release_grad_slots()

    def pack(self, phase: Optional[int] = None):
        """gather the gradients (of one backward phase, or of all) into their views of the flat buffer"""
        srcs, dsts = [], []
        deferred = None
        if self.flat.is_cuda:
            from . import ops
            deferred = ops.deferred_dest
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            if phase is not None and self.phase[i] != phase:
                continue
            if p.grad is None:
                if self._dirty[i]:      # a parameter that got no gradient this step: zero (once; the buffer starts zeroed)
                    v.zero_()
                    self._dirty[i] = False
            else:
                self._dirty[i] = True
                # a slice written by a deferred, grouped launch is already final: what autograd holds for it may be a copy taken
                # BEFORE that launch ran (ops._DEFERRED_DESTS)
                if p.grad.data_ptr() != v.data_ptr() and not (deferred is not None and deferred(v.data_ptr(), v.numel() * 4)):
                    srcs.append(p.grad)
                    dsts.append(v)
            p.grad = v
        if srcs:
            torch._foreach_copy_(dsts, srcs)

    def all_reduce_mean(self, group=None, phase: Optional[int] = None, async_op: bool = False, _force: bool = False):
        """mean over ranks of the whole buffer or of one phase's slice; with async_op returns a handle with .wait()
        (`_force`: issue the collective even in a one-rank group -- tests of the RCCL branch on one-GPU boxes)"""
        if not (dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or _force)):
            return None
        if phase is None:
            buf = self.flat
        elif isinstance(phase, (tuple, list)):          # a run of consecutive phases: one slice (the layout is phase by phase)
            buf = self.flat[self.segments[phase[0]][0]:self.segments[phase[-1]][1]]
        else:
            buf = self.flat[self.segments[phase][0]:self.segments[phase][1]]
        if buf.numel() == 0:
            return None
        if dist.get_backend(group) == "nccl":       # RCCL averages in the collective: no separate scaling pass
            work = dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=group, async_op=async_op)
            return work if async_op else None
        world = dist.get_world_size(group)
        work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if not async_op:
            buf.div_(world)
            return None
        return _ScaledWork(work, buf, world)


class _ScaledWork:
    """async SUM all-reduce + the division that turns it into a mean (backends without ReduceOp.AVG: gloo)"""

    def __init__(self, work, buf, world):
        self.work, self.buf, self.world = work, buf, world

    def wait(self):
        self.work.wait()
        self.buf.div_(self.world)


class FlatAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (reference optimizers.py:196) as ONE HIP kernel over flat buffers.

    The parameters are re-pointed at views of one flat fp32 buffer (values, names and state_dict are unchanged), the
    gradients already live in the FlatGradBucket, the moments are flat: an update is a single streaming pass
    (7 x 4 B per parameter) instead of torch's multi-tensor launches over ~60 tensors.

    It IS a torch.optim.Optimizer with one param_group, so the reference's LR schedulers (optimizers.py:199-245) attach to
    it unchanged; lr / betas / eps / weight_decay are read by the kernel from a 5-float DEVICE buffer that `step()` refreshes
    from `param_groups[0]` whenever they changed -- a captured (hipGraph) update follows the schedule without re-capture."""

    def __init__(self, bucket: "FlatGradBucket", lr: float, weight_decay: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.bucket = bucket
        super().__init__(bucket.model_order, dict(lr=float(lr), betas=tuple(betas), eps=float(eps), weight_decay=float(weight_decay)))
        ps = bucket.params
        with torch.no_grad():       # same layout as the gradient bucket (aligned starts, fused groups back to back)
            self.flat_p = torch.zeros_like(bucket.flat)
            for p, off in zip(ps, bucket.offsets):
                v = self.flat_p[off:off + p.numel()].view_as(p)
                v.copy_(p.detach())
                p.data = v
        self.m = torch.zeros_like(self.flat_p)
        self.v = torch.zeros_like(self.flat_p)
        self.step_count = torch.zeros(1, dtype=torch.float32, device=self.flat_p.device)
        self.hyper = torch.zeros(5, dtype=torch.float32, device=self.flat_p.device)
        self._hyper_host = None
        self.sync_hyper()

    # ---- hyper-parameters live in param_groups[0] (what schedulers write) and are mirrored to the device buffer
    def _g(self):
        return self.param_groups[0]

    lr = property(lambda self: float(self._g()["lr"]), lambda self, v: self._g().__setitem__("lr", float(v)))
    wd = property(lambda self: float(self._g()["weight_decay"]), lambda self, v: self._g().__setitem__("weight_decay", float(v)))
    eps = property(lambda self: float(self._g()["eps"]), lambda self, v: self._g().__setitem__("eps", float(v)))
    betas = property(lambda self: tuple(self._g()["betas"]), lambda self, v: self._g().__setitem__("betas", tuple(v)))

    def set_lr(self, lr: float):
        self.lr = lr

    def sync_hyper(self):
        """upload {lr, beta1, beta2, eps, weight_decay} if they changed since the last upload (NOT capturable: call before replay)"""
        g = self._g()
        cur = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))
        if cur != self._hyper_host:
            self.hyper.copy_(torch.tensor(cur, dtype=torch.float32), non_blocking=False)
            self._hyper_host = cur

    # ---- checkpoint compatibility with torch.optim.AdamW (what the reference's save_ckpt / load_ckpt move around,
    # trainer_utils.py:23-92 with optimizers.py:196): same state_dict layout, parameters indexed in model order
    def _views(self, flat):
        off = {id(p): o for p, o in zip(self.bucket.params, self.bucket.offsets)}
        return [flat[off[id(p)]:off[id(p)] + p.numel()].view_as(p) for p in self.bucket.model_order]

    def state_dict(self):
        order = self.bucket.model_order
        ref = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=self.lr, betas=self.betas, eps=self.eps,
                                weight_decay=self.wd)
        group = dict(ref.param_groups[0])          # every key this torch version expects, with AdamW's defaults
        for k, v in self._g().items():             # scheduler bookkeeping (initial_lr, ...) travels with the group
            if k != "params" and k not in ("lr", "betas", "eps", "weight_decay"):
                group.setdefault(k, v)
        group["params"] = list(range(len(order)))
        step = float(self.step_count.item())
        state = {}
        if step > 0:
            for i, (m, v) in enumerate(zip(self._views(self.m), self._views(self.v))):
                state[i] = {"step": torch.tensor(step), "exp_avg": m.clone(), "exp_avg_sq": v.clone()}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.bucket.model_order):
            raise ValueError("FlatAdamW.load_state_dict: expected one parameter group covering every trainable parameter")
        g = groups[0]
        self.lr, self.wd, self.eps = float(g["lr"]), float(g["weight_decay"]), float(g["eps"])
        self.betas = tuple(float(b) for b in g["betas"])
        if "initial_lr" in g:
            self._g()["initial_lr"] = g["initial_lr"]
        steps = {float(st["step"]) for st in sd["state"].values()}
        if len(steps) > 1:
            raise ValueError("FlatAdamW.load_state_dict: per-parameter step counts differ; the flat update has one counter")
        with torch.no_grad():
            self.m.zero_(); self.v.zero_()
            for i, (m, v) in enumerate(zip(self._views(self.m), self._views(self.v))):
                st = sd["state"].get(i, sd["state"].get(str(i)))
                if st is not None:
                    m.copy_(st["exp_avg"]); v.copy_(st["exp_avg_sq"])
            self.step_count.fill_(steps.pop() if steps else 0.0)
        self.sync_hyper()

    def zero_grad(self, set_to_none: bool = True):
        self.bucket.clear()

    def step(self, closure=None, _sync: bool = True, ticked: bool = False):
        """`ticked`: the step counter has already been advanced for this update (ops.mse_loss_and_grad(tick=self.step_count))"""
        from . import _lib as L
        from . import ops
        from .ops import _p, _stream
        if _sync:
            self.sync_hyper()
        fn = L.load().gaot_adamw_apply_dev if ticked else L.load().gaot_adamw_step_dev
        L.check(fn(_p(self.flat_p), _p(self.bucket.flat), _p(self.m), _p(self.v), self.flat_p.numel(),
                   _p(self.hyper), _p(self.step_count), _stream()), "gaot_adamw_step_dev")
        ops.bump_weights_generation()      # the kernel writes through raw pointers: Parameter._version does not move


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1) for p in module.parameters()])
        dist.broadcast(flat, src=src, group=group)
        off = 0
        for p in module.parameters():
            p.copy_(flat[off:off + p.numel()].view_as(p))
            off += p.numel()


class _Cuts:
    """cut points of a staged backward: `ops.cut(t)` hands the model a detached leaf and remembers (t, leaf); the trainer
    later continues with t.backward(leaf.grad), one phase at a time, newest cut first."""

    def __init__(self):
        self.pairs = []

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        if not (torch.is_grad_enabled() and t.requires_grad):
            return t
        leaf = t.detach().requires_grad_(True)
        self.pairs.append((t, leaf))
        return leaf


class TrainStep:
    """One reference-trainer step on fixed shapes.  `static` holds everything forward() needs except pndata."""

    def __init__(self, model: torch.nn.Module, lr: float = 8e-4, weight_decay: float = 1e-5, use_graph: bool = True,
                 group=None, staged: Optional[bool] = None, stage_groups=None):
        """`staged`: split the backward at the model's cut points and reduce each phase's gradient slice while the next phase
        runs (default: whenever there is more than one rank and the model offers `backward_phases()`)."""
        self.model = model
        model.auto_graph = False          # this harness captures the whole step itself (autograph.py serves plain eager loops)
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        broadcast_parameters(model, 0, group)
        groups = [g for m in model.modules() if hasattr(m, "fused_weight_groups") for g in m.fused_weight_groups()]
        phases = model.backward_phases() if hasattr(model, "backward_phases") else None
        if staged is None:
            staged = self.world > 1
        self.staged = bool(staged and phases is not None and len(phases) > 1)
        self.bucket = FlatGradBucket(list(model.parameters()), groups, phases if self.staged else None)
        # Stage groups: runs of consecutive backward phases that share ONE hipGraph, ONE grouped weight-gradient launch and ONE
        # all-reduce.  A grouped launch lasts as long as one workgroup's K loop however few products it holds, so a launch per
        # phase costs 4 x 140 us where the whole pass takes 265 (measured, bench configuration); the default is therefore two groups:
        # [every phase but the last] and [the last: the encoder], i.e. one all-reduce of the decoder's and the processor's gradients
        # (13.5 of 13.6 MB) that runs while the encoder's backward computes, and a small exposed one.  GAOT_STAGE_GROUPS=each gives
        # one group per phase (round 2's schedule), =pairs two phases per group.
        n_ph = self.bucket.n_phases
        import os as _os
        mode = _os.environ.get("GAOT_STAGE_GROUPS", "default") if stage_groups is None else stage_groups
        if not self.staged:
            self.stage_groups = [[0]]
        elif isinstance(mode, (list, tuple)):
            self.stage_groups = [list(g) for g in mode]
        elif mode == "each":
            self.stage_groups = [[k] for k in range(n_ph)]
        elif mode == "pairs":
            self.stage_groups = [list(range(k, min(k + 2, n_ph))) for k in range(0, n_ph, 2)]
        else:
            self.stage_groups = [list(range(n_ph - 1)), [n_ph - 1]] if n_ph > 1 else [[0]]
        if any(len(g) == 0 for g in self.stage_groups) or [k for g in self.stage_groups for k in g] != list(range(n_ph if self.staged else 1)):
            raise ValueError(f"stage_groups must list the phases 0..{n_ph - 1} in order in non-empty groups, got {self.stage_groups}")
        dev = self.bucket.flat.device
        on_gpu = dev.type == "cuda"
        if on_gpu:
            self.opt = FlatAdamW(self.bucket, lr=lr, weight_decay=weight_decay)
        else:           # CPU (gloo tests of the data-parallel plumbing): torch's own AdamW, parameters in MODEL order
            self.opt = torch.optim.AdamW(self.bucket.model_order, lr=lr, weight_decay=weight_decay)
        # (per-step random neighbour sub-sampling, MAGNOConfig.sampling_strategy, is drawn on the device with a device-resident seed --
        # plan.DropPlan -- so a captured step draws a fresh subset on every replay)
        self.use_graph = use_graph and on_gpu
        self._graphs: Optional[List[torch.cuda.CUDAGraph]] = None
        self._g_opt: Optional[torch.cuda.CUDAGraph] = None
        self._x = self._y = self._loss = None
        self._kwargs: Dict = {}
        self._vx = False              # bound to a vx batch whose geometry may change per step (bind)
        self._vx_unions = None
        self._graph_sets: Dict = {}
        self._cuts: Optional[_Cuts] = None
        self._checked_phases = False
        self._seed = None
        self._ticked = False
        self._merged = False
        self._scratch: Dict = {}      # this step's own tickets / counters / loss partials (ops.scratch_owner)
        self.fused_loss = True        # loss + its gradient + the optimizer tick in one launch (False: ops.mse_loss through autograd)
        self.comm_enabled = True      # False: skip the gradient exchange (bench.py measures the exposed communication as the difference)
        self.force_comm = False       # True: issue the collectives even in a one-rank group (tests of the RCCL path on one-GPU boxes)

    # ---- the eager pieces
    def _forward_loss(self):
        """forward + loss.  On the GPU the loss, its gradient (unit seed) and the optimizer's tick come from ONE launch; the pair
        (pred, dpred) waits in self._seed for _backward_loss()."""
        self.bucket.clear()
        if self._vx_unions is not None:       # vx: the unions' tables were uploaded by _vx_load(); the forward only refreshes (capturable)
            self.model.encoder._vx_preloaded = (self._vx_unions[0], self._kwargs["encoder_nbrs"])
            self.model.decoder._vx_preloaded = (self._vx_unions[1], self._kwargs["decoder_nbrs"])
        pred = self.model(pndata=self._x, **self._kwargs)
        if pred.is_cuda:
            from . import ops
            if self.fused_loss and pred.requires_grad:
                loss, dpred = ops.mse_loss_and_grad(pred, self._y, tick=self.opt.step_count)
                self._seed = (pred, dpred)
                self._ticked = True
                return loss
            return ops.mse_loss(pred, self._y)
        return torch.nn.functional.mse_loss(pred, self._y)          # CPU: gloo tests of the data-parallel plumbing

    def _backward_loss(self, loss):
        seed, self._seed = self._seed, None
        if seed is not None:
            self._backward(*seed)
        else:
            self._backward(loss)

    def _forward_backward(self):
        """unstaged: forward, loss, the whole backward, gradients packed"""
        loss = self._forward_loss()
        self._backward_loss(loss)
        self.bucket.pack()
        return loss.detach()

    @staticmethod
    def _backward(t, grad=None):
        """backward pass with the weight-gradient products deferred to ONE grouped launch at its end (ops.deferred_wgrad)"""
        if t.is_cuda:
            from . import ops
            if grad is None and t.dim() == 0:
                # the seed gradient of the loss: a kept 1.0 instead of the ones_like() fill autograd would launch every step
                one = TrainStep._ONES.get(t.device)
                if one is None:
                    one = TrainStep._ONES[t.device] = torch.ones((), device=t.device, dtype=t.dtype)
                grad = one
            with ops.deferred_wgrad():
                t.backward(grad)
        else:
            t.backward(grad)

    _ONES: Dict = {}

    def _phase0(self):
        """staged: forward with cut points + the backward of phase 0 (everything after the last cut)"""
        from . import ops
        self._cuts = _Cuts()
        ops.set_cut_hook(self._cuts)
        try:
            loss = self._forward_loss()
        finally:
            ops.set_cut_hook(None)
        if len(self._cuts.pairs) != self.bucket.n_phases - 1:
            raise RuntimeError(f"model marked {len(self._cuts.pairs)} cut points but lists {self.bucket.n_phases} backward phases")
        self._backward_loss(loss)
        self._check_phase(0)
        self.bucket.pack(0)
        return loss.detach()

    def _phase(self, k: int):
        t, leaf = self._cuts.pairs[-k]
        if leaf.grad is None:
            raise RuntimeError(f"cut point {len(self._cuts.pairs) - k} received no gradient")
        self._backward(t, leaf.grad)
        self._check_phase(k)
        self.bucket.pack(k)
        if k == self.bucket.n_phases - 1:
            self._cuts = None

    def _run_group(self, gi: int):
        """the phases of one stage group back to back, their weight-gradient products in ONE grouped launch at the group's end"""
        loss = None
        if self.bucket.flat.is_cuda:
            from . import ops
            scope = ops.deferred_wgrad()
        else:
            import contextlib
            scope = contextlib.nullcontext()
        with scope:
            for k in self.stage_groups[gi]:
                if k == 0:
                    loss = self._phase0()
                else:
                    self._phase(k)
        return loss

    def _check_phase(self, k: int):
        """first eager pass only: after phase k no parameter of a LATER phase may hold a gradient yet (the slices reduced
        so far are final) -- guards a model whose backward_phases() and cut points disagree"""
        if self._checked_phases:
            return
        for p, ph in zip(self.bucket.params, self.bucket.phase):
            if ph > k and p.grad is not None:
                raise RuntimeError("backward_phases() disagrees with the cut points: a later-phase parameter already has a gradient")
        if k == self.bucket.n_phases - 1:
            self._checked_phases = True

    def _eager_step(self):
        if self.bucket.flat.is_cuda:
            from . import ops
            with ops.scratch_owner(self._scratch):
                return self._eager_step_inner()
        return self._eager_step_inner()

    def _eager_step_inner(self):
        self._ticked = False          # set by _forward_loss when the loss launch advanced the optimizer's step counter
        if not self.staged:
            loss = self._forward_backward()
            self.bucket.all_reduce_mean(self.group, _force=self.force_comm)
        else:
            works, loss = [], None
            for gi, ks in enumerate(self.stage_groups):
                out = self._run_group(gi)
                loss = out if out is not None else loss
                works.append(self.bucket.all_reduce_mean(self.group, ks, async_op=True, _force=self.force_comm))
            for w in works:
                if w is not None:
                    w.wait()
        if self._ticked:
            self.opt.step(ticked=True)
        else:
            self.opt.step()
        return loss

    def bind(self, pndata: torch.Tensor, target: torch.Tensor, **forward_kwargs):
        """Fix the static buffers (shapes) of the step; later `step()` calls copy new data into them.

        vx (xcoord [B, N, d] with caller-supplied per-sample `encoder_nbrs` / `decoder_nbrs`, static_trainer.py:180-202): the coordinates
        become a static buffer as well and the per-sample graphs may change with every `step(..., xcoord=, encoder_nbrs=, decoder_nbrs=)`:
        the batch's unions live in static padded buffers (plan.StaticUnion) that the captured step re-composes on the device from a small
        table uploaded per step -- one captured step per edge-count bucket serves every batch composition a shuffling loader produces."""
        self._x = pndata.clone()
        self._y = target.clone()
        self._kwargs = dict(forward_kwargs)
        self._graphs = self._g_opt = None
        self._graph_sets = {}
        self._vx_unions = None
        xc = forward_kwargs.get("xcoord")
        self._vx = bool(torch.is_tensor(xc) and xc.dim() == 3 and xc.is_cuda and forward_kwargs.get("encoder_nbrs") is not None
                        and forward_kwargs.get("decoder_nbrs") is not None and forward_kwargs.get("query_coord") is None
                        and all(hasattr(m, "vx_static_ok") for m in (getattr(self.model, "encoder", None), getattr(self.model, "decoder", None)))
                        and self._vx_sides_ok())
        if self._vx:
            self._kwargs["xcoord"] = xc.detach().clone().contiguous()
            self._kwargs["latent_tokens_coord"] = forward_kwargs["latent_tokens_coord"].detach().contiguous()

    def _vx_sides_ok(self) -> bool:
        from . import plan as P
        return P.VX_STATIC and all(bool(m.precompute_edges) and not m.node_embedding for m in (self.model.encoder, self.model.decoder))

    MAX_GRAPH_SETS = 6      # vx: captured steps kept (one per combination of the unions' edge buckets), least recently used first out

    def _vx_load(self):
        """upload this step's union tables (per scale, encoder and decoder) and pick the captured step of their buckets"""
        kw = self._kwargs
        B = self._x.shape[0]
        enc = self.model.encoder.vx_unions(kw["encoder_nbrs"], kw["xcoord"], kw["latent_tokens_coord"], B)
        dec = self.model.decoder.vx_unions(kw["decoder_nbrs"], kw["latent_tokens_coord"], kw["xcoord"], B)
        self._vx_unions = (enc, dec)
        key = tuple((u.uid, u.pending_raw) for u in enc + dec)          # (raw lists or per-sample plans: other launches)
        hit = self._graph_sets.pop(key, None)
        if hit is None:
            while len(self._graph_sets) >= self.MAX_GRAPH_SETS:
                self._graph_sets.pop(next(iter(self._graph_sets)))
            hit = {"graphs": None, "g_opt": None, "loss": None, "merged": False, "ticked": False, "unions": (enc, dec)}
        self._graph_sets[key] = hit
        return hit

    def _capture(self):
        from . import ops
        with ops.scratch_owner(self._scratch):
            self._capture_inner()

    def _capture_inner(self):
        # warm-up iterations must not advance training: snapshot weights + optimizer state, restore after capture
        snap = [t.clone() for t in (self.opt.flat_p, self.opt.m, self.opt.v, self.opt.step_count)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up: builds GeometryPlans (they sync once), fills the allocator
            for _ in range(2):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # thread_local: the RCCL watchdog thread's event queries must not invalidate the capture
        pool = torch.cuda.graph_pool_handle()
        self._graphs = []
        self._ticked = False
        # one rank, nothing to exchange between backward and update: the optimizer joins the step's graph (one replay per step)
        self._merged = self.world == 1 and not self.force_comm and not self.staged
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
            self._loss = self._run_group(0) if self.staged else self._forward_backward()
            if self._merged:
                self.opt.step(_sync=False, ticked=self._ticked)
        self._graphs.append(g)
        for gi in range(1, len(self.stage_groups) if self.staged else 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                self._run_group(gi)
            self._graphs.append(g)
        self._g_opt = None
        if not self._merged:
            self._g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._g_opt, pool=pool, capture_error_mode="thread_local"):
                self.opt.step(_sync=False, ticked=self._ticked)          # (graphs exist on the GPU only: FlatAdamW)
        for dst, src in zip((self.opt.flat_p, self.opt.m, self.opt.v, self.opt.step_count), snap):
            dst.copy_(src)

    def step(self, pndata: Optional[torch.Tensor] = None, target: Optional[torch.Tensor] = None, xcoord: Optional[torch.Tensor] = None,
             encoder_nbrs=None, decoder_nbrs=None) -> torch.Tensor:
        """one training step on new fields and -- vx -- a new geometry: coordinates [B, N, d] and the per-sample neighbour lists of this batch
        (what static_trainer.py:180-202 hands the model every step), any composition of samples of the bound shapes"""
        if pndata is not None:
            self._x.copy_(pndata, non_blocking=True)
        if target is not None:
            self._y.copy_(target, non_blocking=True)
        if xcoord is not None or encoder_nbrs is not None or decoder_nbrs is not None:
            if not self._vx:
                raise RuntimeError("TrainStep.step: a new geometry per step needs a vx binding (xcoord [B, N, d] with encoder_nbrs / decoder_nbrs, "
                                   "no node_embedding); bind() again for another fixed geometry")
            if xcoord is not None:
                self._kwargs["xcoord"].copy_(xcoord, non_blocking=True)
            if encoder_nbrs is not None:
                self._kwargs["encoder_nbrs"] = encoder_nbrs
            if decoder_nbrs is not None:
                self._kwargs["decoder_nbrs"] = decoder_nbrs
        return self._step()

    def _step(self) -> torch.Tensor:
        gs = None
        if self._vx:
            gs = self._vx_load()
            self._graphs, self._g_opt, self._loss, self._merged, self._ticked = gs["graphs"], gs["g_opt"], gs["loss"], gs["merged"], gs["ticked"]
        if not self.use_graph:
            return self._eager_step()
        if self._graphs is not None and self._merged and self.force_comm:
            self._graphs = None                  # the exchange was switched on after a one-graph capture: split the step again
        if self._graphs is None:
            self._capture()
            if gs is not None:
                gs.update(graphs=self._graphs, g_opt=self._g_opt, loss=self._loss, merged=self._merged, ticked=self._ticked)
        self.opt.sync_hyper()                    # a scheduler may have moved lr since the last step (device buffer, no re-capture)
        if not self.staged:
            self._graphs[0].replay()
            if self.comm_enabled:
                self.bucket.all_reduce_mean(self.group, _force=self.force_comm)      # one flat RCCL all-reduce between the two graphs
        else:
            # everything below is enqueued without a host wait: a stage group's slice is reduced on RCCL's stream while the main
            # stream replays the backward of the next group
            works = []
            for gi, g in enumerate(self._graphs):
                g.replay()
                if self.comm_enabled:
                    works.append(self.bucket.all_reduce_mean(self.group, self.stage_groups[gi], async_op=True, _force=self.force_comm))
            for w in works:
                if w is not None:
                    w.wait()
        if self._g_opt is not None:
            self._g_opt.replay()
        from . import ops
        ops.bump_weights_generation()
        return self._loss
