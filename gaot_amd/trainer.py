"""Training-step harness with the semantics of the reference trainer step
(static_trainer.py:160-178, optimizers.py:247-257, loss base_trainer.py:71):

    zero_grad -> pred = model(...) -> MSELoss(mean) -> backward -> AdamW.step

plus what the reference lacks: data-parallel training.  One process per GPU; all parameter
gradients live in ONE flat fp32 buffer (param.grad are views), so the exchange is a single RCCL all-reduce
(13.6 MB at the example config) per step; parameters are broadcast from rank 0 at start (the reference seeds
with seed+rank and never synchronises).  With static shapes the forward+backward and the optimizer update are
captured as hipGraphs (geometry-only work is cached in GeometryPlans and stays outside the graph).
"""
from typing import Dict, List, Optional

import torch
import torch.distributed as dist


def shard_indices(n_samples: int, rank: int, world: int, epoch: int = 0, shuffle: bool = True, seed: int = 0) -> List[int]:
    """DistributedSampler-equivalent: same permutation on every rank, padded to a multiple of `world`, strided."""
    g = torch.Generator().manual_seed(seed + epoch)
    order = torch.randperm(n_samples, generator=g).tolist() if shuffle else list(range(n_samples))
    total = ((n_samples + world - 1) // world) * world
    order += order[: total - n_samples]
    return order[rank:total:world]


class FlatGradBucket:
    """All gradients in one contiguous fp32 buffer (one RCCL all-reduce per step, one fused optimizer pass).

    Autograd is left to hand over freshly produced gradient tensors (param.grad is None before backward, so
    AccumulateGrad steals instead of launching one add per parameter); `pack()` then gathers them with ONE
    multi-tensor copy and re-points param.grad at views of the flat buffer for the all-reduce / optimizer."""

    ALIGN = 64      # floats: every parameter (or fused group) starts on a 256-byte boundary -> 16-byte vector kernels apply

    def __init__(self, params: List[torch.nn.Parameter], groups: Optional[List[List[torch.nn.Parameter]]] = None):
        """`groups`: lists of parameters that must sit back to back (in the given order, no padding) so that a fused
        GEMM can read them as ONE matrix (q|k|v, w1|w3: ops.adjacent_rows) instead of concatenating every step."""
        params = [p for p in params if p.requires_grad]
        self.model_order = list(params)           # the order torch optimizers index parameters by (checkpoint compatibility)
        pos = {id(p): i for i, p in enumerate(params)}
        follow, skip = {}, set()
        for g in groups or []:
            if len(g) > 1 and all(id(p) in pos for p in g) and not any(id(p) in skip or id(p) in follow for p in g):
                follow[id(g[0])] = list(g[1:])
                skip.update(id(p) for p in g[1:])
        self.params, self.offsets = [], []
        off = 0
        for p in params:
            if id(p) in skip:
                continue
            off = -(-off // self.ALIGN) * self.ALIGN
            for q in [p] + follow.get(id(p), []):
                self.params.append(q)
                self.offsets.append(off)
                off += q.numel()
        self.numel = -(-off // 4) * 4
        ref = self.params[0]
        self.flat = torch.zeros(self.numel, dtype=ref.dtype, device=ref.device)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for p, o in zip(self.params, self.offsets)]
        self._dirty = [False] * len(self.params)      # view holds a gradient from an earlier step

    def clear(self):
        for p in self.params:
            p.grad = None
        if self.flat.is_cuda:         # weight-gradient GEMMs of the HIP ops write straight into their views (ops._claim)
            from . import ops
            if ops._GRAD_SLOTS.get(id(self.params[0]), [None])[0] is not self.views[0]:
                ops.register_grad_slots(self.params, self.views)
            ops.release_grad_slots()

    def pack(self):
        srcs, dsts = [], []
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            if p.grad is None:
                if self._dirty[i]:      # a parameter that got no gradient this step: zero (once; the buffer starts zeroed)
                    v.zero_()
                    self._dirty[i] = False
            else:
                self._dirty[i] = True
                if p.grad.data_ptr() != v.data_ptr():
                    srcs.append(p.grad)
                    dsts.append(v)
        if srcs:
            torch._foreach_copy_(dsts, srcs)
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce_mean(self, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            if dist.get_backend(group) == "nccl":       # RCCL averages in the collective: no separate scaling pass
                dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=group)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
                self.flat.div_(dist.get_world_size(group))


class FlatAdamW:
    """torch.optim.AdamW semantics (reference optimizers.py:196) as ONE HIP kernel over flat buffers.

    The parameters are re-pointed at views of one flat fp32 buffer (values, names and state_dict are unchanged), the
    gradients already live in the FlatGradBucket, the moments are flat: an update is a single streaming pass
    (7 x 4 B per parameter) instead of torch's multi-tensor launches over ~60 tensors."""

    def __init__(self, bucket: "FlatGradBucket", lr: float, weight_decay: float, betas=(0.9, 0.999), eps: float = 1e-8):
        self.bucket = bucket
        self.lr, self.wd, self.betas, self.eps = float(lr), float(weight_decay), betas, float(eps)
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

    def step(self):
        from . import _lib as L
        from .ops import _p, _stream
        L.check(L.load().gaot_adamw_step(_p(self.flat_p), _p(self.bucket.flat), _p(self.m), _p(self.v), self.flat_p.numel(),
                                         self.lr, self.betas[0], self.betas[1], self.eps, self.wd, _p(self.step_count), _stream()),
                "gaot_adamw_step")


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


class TrainStep:
    """One reference-trainer step on fixed shapes.  `static` holds everything forward() needs except pndata."""

    def __init__(self, model: torch.nn.Module, lr: float = 8e-4, weight_decay: float = 1e-5, use_graph: bool = True,
                 group=None):
        self.model = model
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        broadcast_parameters(model, 0, group)
        groups = [g for m in model.modules() if hasattr(m, "fused_weight_groups") for g in m.fused_weight_groups()]
        self.bucket = FlatGradBucket(list(model.parameters()), groups)
        dev = self.bucket.flat.device
        on_gpu = dev.type == "cuda"
        if on_gpu:
            self.opt = FlatAdamW(self.bucket, lr=lr, weight_decay=weight_decay)
        else:           # CPU (gloo tests of the data-parallel plumbing): torch's own AdamW
            self.opt = torch.optim.AdamW(self.bucket.params, lr=lr, weight_decay=weight_decay)
        self.use_graph = use_graph and on_gpu
        self._g_fb: Optional[torch.cuda.CUDAGraph] = None
        self._g_opt: Optional[torch.cuda.CUDAGraph] = None
        self._x = self._y = self._loss = None
        self._kwargs: Dict = {}

    # ---- the eager pieces
    def _forward_backward(self):
        self.bucket.clear()
        pred = self.model(pndata=self._x, **self._kwargs)
        if pred.is_cuda:
            from . import ops
            loss = ops.mse_loss(pred, self._y)
        else:           # CPU: gloo tests of the data-parallel plumbing
            loss = torch.nn.functional.mse_loss(pred, self._y)
        loss.backward()
        self.bucket.pack()
        return loss.detach()

    def bind(self, pndata: torch.Tensor, target: torch.Tensor, **forward_kwargs):
        """Fix the static buffers (shapes) of the step; later `step()` calls copy new data into them."""
        self._x = pndata.clone()
        self._y = target.clone()
        self._kwargs = forward_kwargs
        self._g_fb = self._g_opt = None

    def _capture(self):
        # warm-up iterations must not advance training: snapshot weights + optimizer state, restore after capture
        snap = [t.clone() for t in (self.opt.flat_p, self.opt.m, self.opt.v, self.opt.step_count)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up: builds GeometryPlans (they sync once), fills the allocator
            for _ in range(2):
                self._forward_backward()
                self.bucket.all_reduce_mean(self.group)
                self.opt.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._g_fb = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread's event queries must not invalidate the capture
        with torch.cuda.graph(self._g_fb, capture_error_mode="thread_local"):
            self._loss = self._forward_backward()
        self._g_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._g_opt, capture_error_mode="thread_local"):
            self.opt.step()
        for dst, src in zip((self.opt.flat_p, self.opt.m, self.opt.v, self.opt.step_count), snap):
            dst.copy_(src)

    def step(self, pndata: Optional[torch.Tensor] = None, target: Optional[torch.Tensor] = None) -> torch.Tensor:
        if pndata is not None:
            self._x.copy_(pndata, non_blocking=True)
        if target is not None:
            self._y.copy_(target, non_blocking=True)
        if self.use_graph:
            if self._g_fb is None:
                self._capture()
            self._g_fb.replay()
            self.bucket.all_reduce_mean(self.group)      # one flat RCCL all-reduce between the two graphs
            self._g_opt.replay()
            return self._loss
        loss = self._forward_backward()
        self.bucket.all_reduce_mean(self.group)
        self.opt.step()
        return loss
