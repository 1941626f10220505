"""Data-parallel step on CPU with gloo, world_size 2 (the N > 1 path of bench.py minus the GPU kernels):
parameter broadcast from rank 0, flat gradient bucket, one all-reduce(mean), identical AdamW updates, and
equivalence with single-process training on the concatenated global batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gaot_amd.trainer import FlatGradBucket, TrainStep, broadcast_parameters, shard_indices


class TinyOperator(torch.nn.Module):
    """stand-in with GAOT's call convention (keyword `pndata`); the comm layer is model agnostic"""

    def __init__(self, seed):
        super().__init__()
        torch.manual_seed(seed)
        self.a = torch.nn.Linear(3, 16)
        self.b = torch.nn.Linear(16, 2, bias=False)
        self.unused = torch.nn.Parameter(torch.ones(4))     # a parameter that never receives a gradient

    def forward(self, pndata, scale=1.0):
        from gaot_amd import ops
        h = ops.cut(torch.tanh(self.a(pndata)))             # cut point of the staged backward (identity without a hook)
        return self.b(h) * scale

    def backward_phases(self):
        return [[self.b.weight], [self.a.weight, self.a.bias, self.unused]]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, staged=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = TinyOperator(seed=100 + rank)             # reference behaviour: seed + rank -> ranks start DIFFERENT
    ts = TrainStep(model, lr=1e-2, weight_decay=1e-3, staged=staged)
    assert ts.staged == (staged is not False) and ts.bucket.n_phases == (2 if ts.staged else 1)
    g = torch.Generator().manual_seed(7)
    x_all, y_all = torch.randn(8, 5, 3, generator=g), torch.randn(8, 5, 2, generator=g)
    idx = shard_indices(8, rank, world, shuffle=False)
    ts.bind(x_all[idx], y_all[idx], scale=0.5)
    losses = [float(ts.step()) for _ in range(3)]
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        out.put((gathered[0].tolist(), gathered[1].tolist(), losses))     # plain lists: no shared-memory handles
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("staged", [None, False])
def test_two_rank_gloo_step_matches_single_process(staged):
    """staged=None: the default for > 1 rank -- backward split at the cut point, one async all-reduce per phase slice;
    staged=False: one all-reduce of the whole flat buffer after the backward pass.  Both equal single-process training."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, staged)) for r in range(2)]
    for p in procs:
        p.start()
    p0, p1, _ = q.get(timeout=120)
    p0, p1 = torch.tensor(p0), torch.tensor(p1)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(p0, p1)                        # ranks stay bit-identical after the all-reduced updates
    # single process on the global batch: mean of per-rank mean-losses == global mean loss (equal shard sizes)
    model = TinyOperator(seed=100)                    # rank 0's weights are what got broadcast
    ts = TrainStep(model, lr=1e-2, weight_decay=1e-3)
    g = torch.Generator().manual_seed(7)
    x_all, y_all = torch.randn(8, 5, 3, generator=g), torch.randn(8, 5, 2, generator=g)
    ts.bind(x_all, y_all, scale=0.5)
    for _ in range(3):
        ts.step()
    ref = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    assert torch.allclose(p0, ref, rtol=1e-5, atol=1e-6)


def test_shard_indices_partition():
    for n, w in [(10, 2), (7, 2), (16, 8), (5, 8), (3, 8), (1, 4)]:
        shards = [shard_indices(n, r, w, epoch=3, seed=1) for r in range(w)]
        assert len({len(s) for s in shards}) == 1
        assert set(i for s in shards for i in s) == set(range(n))
        assert shard_indices(n, 0, w, epoch=3, seed=1) == shards[0]
        assert shard_indices(n, 0, w, epoch=4, seed=1) != shards[0] or n <= 2


def test_flat_bucket_views_and_missing_grads():
    m = TinyOperator(0)
    b = FlatGradBucket(list(m.parameters()))
    b.clear()
    m(pndata=torch.randn(2, 3)).sum().backward()
    grads = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    b.pack()
    names = {id(p): n for n, p in m.named_parameters()}
    assert [id(p) for p in b.params] == [id(p) for p in m.parameters()]        # no fused groups: model order
    covered = torch.zeros(b.flat.numel(), dtype=torch.bool)
    for p, off in zip(b.params, b.offsets):
        assert off % FlatGradBucket.ALIGN == 0                                # every parameter starts on a 256-byte boundary
        seg = b.flat[off:off + p.numel()].view_as(p)
        n = names[id(p)]
        assert p.grad.data_ptr() == seg.data_ptr()
        assert torch.equal(seg, grads[n]) if n in grads else bool((seg == 0).all())
        assert not covered[off:off + p.numel()].any()
        covered[off:off + p.numel()] = True
    assert bool((b.flat[~covered] == 0).all())                                 # padding stays zero
    broadcast_parameters(m)      # no process group: a no-op


def test_flat_bucket_fused_groups_are_back_to_back():
    """parameters named by fused_weight_groups() sit contiguously (q|k|v, w1|w3) so ops.adjacent_rows can view them as
    one matrix; everything else keeps model order."""
    from gaot_amd import ops
    lin = [torch.nn.Linear(8, n, bias=False) for n in (4, 12, 4, 8)]
    extra = torch.nn.Parameter(torch.zeros(3))
    params = [lin[0].weight, extra, lin[1].weight, lin[2].weight, lin[3].weight]
    b = FlatGradBucket(params, groups=[[lin[0].weight, lin[2].weight, lin[3].weight]])
    assert [id(p) for p in b.params] == [id(x) for x in (lin[0].weight, lin[2].weight, lin[3].weight, extra, lin[1].weight)]
    o = dict(zip(map(id, b.params), b.offsets))
    assert o[id(lin[2].weight)] == o[id(lin[0].weight)] + 32 and o[id(lin[3].weight)] == o[id(lin[2].weight)] + 32
    # re-point the parameters at a flat buffer with that layout (what FlatAdamW does) and view the group as one matrix
    flat = torch.arange(b.numel, dtype=torch.float32)
    for p, off in zip(b.params, b.offsets):
        p.data = flat[off:off + p.numel()].view_as(p)
    W = ops.adjacent_rows([lin[0].weight, lin[2].weight, lin[3].weight])
    assert W is not None and W.shape == (16, 8) and W.data_ptr() == lin[0].weight.data_ptr()
    assert torch.equal(W, torch.cat([lin[0].weight, lin[2].weight, lin[3].weight]).detach())
    assert ops.adjacent_rows([lin[0].weight, lin[1].weight]) is None          # not adjacent: caller concatenates
    assert torch.equal(ops.stacked_rows([lin[0].weight, lin[1].weight]), torch.cat([lin[0].weight, lin[1].weight]).detach())


def test_flat_bucket_phase_layout():
    """backward phases get contiguous slices in completion order; fused groups stay back to back inside their phase"""
    lin = [torch.nn.Linear(8, n, bias=False) for n in (4, 12, 4, 8)]
    params = [l.weight for l in lin]
    b = FlatGradBucket(params, groups=[[lin[0].weight, lin[2].weight]], phases=[[lin[3].weight, lin[1].weight], [lin[0].weight, lin[2].weight]])
    assert [id(p) for p in b.params] == [id(lin[i].weight) for i in (1, 3, 0, 2)]       # phase 0 in model order, then phase 1
    assert b.phase == [0, 0, 1, 1] and b.n_phases == 2
    (s0, e0), (s1, e1) = b.segments
    assert s0 == 0 and e0 <= s1 and s1 % FlatGradBucket.ALIGN == 0 and e1 == b.numel
    for p, off, ph in zip(b.params, b.offsets, b.phase):
        s_, e_ = b.segments[ph]
        assert s_ <= off and off + p.numel() <= e_
    assert b.model_order == params


def test_staged_backward_equals_plain_backward_single_process():
    """the cut-point machinery alone (no process group): phase-by-phase backward gives the same update as one backward"""
    xs, ys = torch.randn(4, 5, 3), torch.randn(4, 5, 2)
    out = []
    for staged in (False, True):
        m = TinyOperator(3)
        ts = TrainStep(m, lr=1e-2, weight_decay=1e-3, staged=staged)
        assert ts.staged == staged
        ts.bind(xs, ys, scale=2.0)
        for _ in range(2):
            ts.step()
        out.append(torch.cat([p.detach().reshape(-1) for p in m.parameters()]))
    assert torch.equal(out[0], out[1])


class ThreePhaseOperator(torch.nn.Module):
    """three backward phases (two cut points): exercises stage groups that merge phases"""

    def __init__(self, seed):
        super().__init__()
        torch.manual_seed(seed)
        self.a, self.b, self.c = torch.nn.Linear(3, 8), torch.nn.Linear(8, 8), torch.nn.Linear(8, 2)

    def forward(self, pndata):
        from gaot_amd import ops
        h = ops.cut(torch.tanh(self.a(pndata)))
        h = ops.cut(torch.tanh(self.b(h)))
        return self.c(h)

    def backward_phases(self):
        return [list(self.c.parameters()), list(self.b.parameters()), list(self.a.parameters())]


@pytest.mark.parametrize("groups,expect", [(None, [[0, 1], [2]]), ("each", [[0], [1], [2]]), ("pairs", [[0, 1], [2]]), ([[0], [1, 2]], [[0], [1, 2]])])
def test_stage_groups_merge_phases_without_changing_the_update(groups, expect):
    """TrainStep(stage_groups=...): runs of consecutive phases share one backward scope and one all-reduce slice; every grouping
    gives the update of the plain backward.  Default: every phase but the last | the last."""
    xs, ys = torch.randn(4, 5, 3), torch.randn(4, 5, 2)
    ref = ThreePhaseOperator(7)
    t0 = TrainStep(ref, lr=1e-2, weight_decay=1e-3, staged=False)
    t0.bind(xs, ys)
    m = ThreePhaseOperator(7)
    ts = TrainStep(m, lr=1e-2, weight_decay=1e-3, staged=True, stage_groups=groups)
    assert ts.stage_groups == expect and ts.bucket.n_phases == 3
    ts.bind(xs, ys)
    for _ in range(2):
        t0.step()
        ts.step()
    assert all(torch.equal(p, q) for p, q in zip(ref.parameters(), m.parameters()))
    # a group's slice is the contiguous range of its phases
    for g in ts.stage_groups:
        lo, hi = ts.bucket.segments[g[0]][0], ts.bucket.segments[g[-1]][1]
        assert 0 <= lo < hi <= ts.bucket.numel
    with pytest.raises(ValueError):
        TrainStep(ThreePhaseOperator(7), staged=True, stage_groups=[[0, 2], [1]])


class GatedThreePhaseOperator(ThreePhaseOperator):
    """as ThreePhaseOperator, plus a parameter (`gate`) that only enters the output when the rank's shard has a positive mean: on the
    other ranks it receives NO gradient in that step (param.grad stays None there; the flat bucket must contribute zeros)"""

    def __init__(self, seed):
        super().__init__(seed)
        self.gate = torch.nn.Parameter(torch.full((2,), 0.5))

    def forward(self, pndata):
        y = super().forward(pndata)
        return y + self.gate * pndata[..., :2] if float(pndata.mean()) > 0 else y

    def backward_phases(self):
        return [list(self.c.parameters()) + [self.gate], list(self.b.parameters()), list(self.a.parameters())]


def _data(n):
    g = torch.Generator().manual_seed(11)
    x, y = torch.randn(n, 5, 3, generator=g), torch.randn(n, 5, 2, generator=g)
    x[::3] += 0.8            # some shards with a positive mean, some without
    return x, y


def _worker_many(rank, world, port, out, n_samples, groups):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = GatedThreePhaseOperator(seed=300 + rank)
    ts = TrainStep(model, lr=1e-2, weight_decay=1e-3, stage_groups=groups)
    assert ts.staged and ts.world == world
    x_all, y_all = _data(n_samples)
    got_gate = []
    for epoch in range(3):
        idx = shard_indices(n_samples, rank, world, epoch=epoch, seed=5)          # world does not divide n_samples: padded by repeats
        ts.bind(x_all[idx], y_all[idx])
        ts.step()
        got_gate.append(float(x_all[idx].mean()) > 0)
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    flags = [None] * world
    dist.all_gather_object(flags, got_gate)
    if rank == 0:
        out.put(([t.tolist() for t in gathered], flags))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_samples,groups", [(4, 10, None), (8, 11, "each"), (8, 5, [[0], [1, 2]])])
def test_many_rank_gloo_staged_step_with_ragged_shards_and_missing_gradients(world, n_samples, groups):
    """world 4 and 8 over gloo, the staged TrainStep with stage groups: the sample count is not a multiple of the world size (and at
    world 8 / 5 samples smaller than it), and `gate` gets no gradient on the ranks whose shard never uses it.  Every rank must end with
    the same weights, equal to a single-process emulation of what DDP means: per-rank gradients (zeros where a rank produced none),
    averaged, one AdamW update."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_many, args=(r, world, port, q, n_samples, groups)) for r in range(world)]
    for p in procs:
        p.start()
    weights, flags = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    weights = [torch.tensor(w) for w in weights]
    assert all(torch.equal(weights[0], w) for w in weights[1:])                    # ranks bit-identical
    assert any(not all(f) for f in flags) and any(any(f) for f in flags)           # the gate really was missing on some ranks, present on others
    model = GatedThreePhaseOperator(seed=300)                                      # rank 0's weights are what got broadcast
    params = list(model.parameters())
    opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=1e-3)
    x_all, y_all = _data(n_samples)
    for epoch in range(3):
        acc = [torch.zeros_like(p) for p in params]
        for r in range(world):
            idx = shard_indices(n_samples, r, world, epoch=epoch, seed=5)
            for p in params:
                p.grad = None
            torch.nn.functional.mse_loss(model(pndata=x_all[idx]), y_all[idx]).backward()
            for a, p in zip(acc, params):
                if p.grad is not None:
                    a += p.grad
        for a, p in zip(acc, params):
            p.grad = a / world
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in params])
    # AdamW's first steps move every entry by ~lr * sign(g): entries whose averaged gradient is rounding noise differ with the
    # summation order of the all-reduce (measured 5e-5); a mishandled shard or a lost gradient would show up at lr = 1e-2
    assert torch.allclose(weights[0], ref, rtol=1e-4, atol=2e-4), float((weights[0] - ref).abs().max())
