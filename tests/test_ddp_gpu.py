"""-m gpu: the data-parallel step with GAOT ITSELF under more than one rank.

A GPU box of the test pool has ONE device, so both ranks run on cuda:0 and exchange gradients over gloo (RCCL refuses two
ranks on one device); everything else is the N > 1 path of bench.py: parameter broadcast, phase-ordered flat gradient
bucket, staged backward with one asynchronous all-reduce per phase slice, hipGraph replay per phase, flat HIP AdamW.
Checked: ranks end bit-identical, and equal to single-process training on the concatenated global batch.
The RCCL branch (ReduceOp.AVG on the flat buffer) is exercised with a one-rank nccl group."""
import os
import socket
from types import SimpleNamespace as NS

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(seed):
    from gaot_amd.model.gaot import GAOT
    from gaot_amd.model.layers.magno import MAGNOConfig
    from gaot_amd.model.layers.attn import TransformerConfig, AttentionConfig
    torch.manual_seed(seed)
    mcfg = MAGNOConfig(coord_dim=2, radius=0.08, hidden_size=64, mlp_layers=3, lifting_channels=32)
    tcfg = TransformerConfig(patch_size=2, hidden_size=128, attn_config=AttentionConfig(num_heads=4, num_kv_heads=4))
    return GAOT(2, 1, NS(args=NS(magno=mcfg, transformer=tcfg), latent_tokens_size=[32, 32]))


def _data():
    g = torch.Generator().manual_seed(21)
    ax = torch.linspace(-1, 1, 32)
    lat = torch.stack(torch.meshgrid(ax, ax, indexing="ij"), -1).reshape(-1, 2)
    x = torch.rand(1500, 2, generator=g) * 2 - 1
    return lat, x, torch.randn(4, 1500, 2, generator=g), torch.randn(4, 1500, 1, generator=g)


def _flat(model):
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()])


def _worker(rank, world, port, out, graph, staged, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    # gloo: every rank on cuda:0 (one-GPU boxes);  nccl (= RCCL): one device per rank
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from gaot_amd.trainer import TrainStep, shard_indices
    model = _build(seed=300 + rank).to(dev).train()           # ranks start DIFFERENT (reference: seed + rank); rank 0 is broadcast
    lat, x, p, t = _data()
    idx = shard_indices(4, rank, world, shuffle=False)
    ts = TrainStep(model, lr=2e-3, weight_decay=1e-4, use_graph=graph, staged=staged)
    assert ts.staged == (staged is not False)
    if ts.staged:                                             # 3 blocks + encoder -> 4 backward phases, 4 slices of the flat buffer
        assert ts.bucket.n_phases == 4 and ts.bucket.segments[-1][1] == ts.bucket.numel
    ts.bind(p[idx].to(dev), t[idx].to(dev), latent_tokens_coord=lat.to(dev), xcoord=x.to(dev))
    losses = [float(ts.step()) for _ in range(3)]
    torch.cuda.synchronize()
    flat = _flat(model).cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        out.put((gathered[0].tolist(), gathered[1].tolist(), losses))
    dist.barrier()
    dist.destroy_process_group()


def _two_ranks(graph, staged, backend):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, graph, staged, backend)) for r in range(2)]
    for p in procs:
        p.start()
    p0, p1, losses = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    p0, p1 = torch.tensor(p0), torch.tensor(p1)
    assert torch.equal(p0, p1)                                # identical after the all-reduced updates
    from gaot_amd.trainer import TrainStep
    dev = torch.device("cuda:0")
    model = _build(seed=300).to(dev).train()
    lat, x, p, t = _data()
    ts = TrainStep(model, lr=2e-3, weight_decay=1e-4, use_graph=False)
    assert ts.staged is False and ts.bucket.n_phases == 1
    ts.bind(p.to(dev), t.to(dev), latent_tokens_coord=lat.to(dev), xcoord=x.to(dev))
    for _ in range(3):
        ts.step()
    ref = _flat(model).cpu()
    # 3 AdamW steps of 2e-3 each: the weights moved by ~6e-3 per entry; agreement to 2e-5 absolute = gradients agree
    assert float((p0 - ref).abs().max()) < 2e-5, float((p0 - ref).abs().max())


@pytest.mark.parametrize("graph,staged", [(True, None), (False, None), (True, False)])
def test_gaot_two_ranks_one_gpu_equals_single_process_global_batch(graph, staged):
    _two_ranks(graph, staged, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("graph,staged", [(True, None), (False, None)])
def test_gaot_two_ranks_two_gpus_over_rccl(graph, staged):
    """the same check with one device per rank and the gradient exchange on RCCL (ReduceOp.AVG, async per phase slice): ranks end
    bit-identical and equal to single-process training on the global batch.  Skipped on one-GPU boxes."""
    _two_ranks(graph, staged, "nccl")


# ---- vx under two ranks: every rank composes ITS OWN shard of a shuffled global batch (static padded unions, one captured step per edge bucket)
def _vx_data():
    from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch
    g = torch.Generator().manual_seed(33)
    ax = torch.linspace(-1, 1, 32)
    lat = torch.stack(torch.meshgrid(ax, ax, indexing="ij"), -1).reshape(-1, 2)
    xs = torch.rand(8, 1500, 2, generator=g) * 2 - 1                     # a dataset of 8 meshes
    return lat, xs, torch.randn(8, 1500, 2, generator=g), torch.randn(8, 1500, 1, generator=g)


def _vx_model(seed):
    m = _build(seed)
    m.encoder.precompute_edges = m.decoder.precompute_edges = True
    return m


_VX_BATCHES = [[0, 5, 2, 7], [3, 1, 6, 4], [7, 2, 0, 3]]                 # three global batches of 4, every one another composition


def _vx_graphs(lat, xs, dev):
    from gaot_amd.model.layers.utils.neighbor_search import NeighborSearch
    ns = NeighborSearch("native")
    latd, xd = lat.to(dev), xs.to(dev)
    return latd, xd, [[ns(xd[i], latd, 0.08)] for i in range(8)], [[ns(latd, xd[i], 0.08)] for i in range(8)]


def _vx_worker(rank, world, port, out, graph):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gaot_amd.trainer import TrainStep
    model = _vx_model(seed=300 + rank).to(dev).train()
    lat, xs, p, t = _vx_data()
    latd, xd, enc, dec = _vx_graphs(lat, xs, dev)
    ts = TrainStep(model, lr=2e-3, weight_decay=1e-4, use_graph=graph)
    mine = lambda b: b[rank::world]                                       # DistributedSampler-style strided shard of the global batch
    b = mine(_VX_BATCHES[0])
    ts.bind(p[b].to(dev), t[b].to(dev), latent_tokens_coord=latd, xcoord=xd[b], encoder_nbrs=[enc[i] for i in b], decoder_nbrs=[dec[i] for i in b])
    for gb in _VX_BATCHES:
        b = mine(gb)
        ts.step(p[b].to(dev), t[b].to(dev), xcoord=xd[b], encoder_nbrs=[enc[i] for i in b], decoder_nbrs=[dec[i] for i in b])
    torch.cuda.synchronize()
    assert ts._vx and (ts._graphs is not None) == graph
    flat = _flat(model).cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        out.put((gathered[0].tolist(), gathered[1].tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("graph", [True, False])
def test_gaot_vx_two_ranks_shuffled_batches_equal_single_process_global_batch(graph):
    """vx training sharded over two ranks (both on cuda:0, gloo): three shuffled global batches of four meshes, every rank composing its own
    two samples per step on static padded unions (hipGraph replay per stage group when `graph`), gradients averaged -- ranks end bit-identical
    and equal to single-process training on the global batches."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_vx_worker, args=(r, 2, port, q, graph)) for r in range(2)]
    for p_ in procs:
        p_.start()
    p0, p1 = q.get(timeout=600)
    for p_ in procs:
        p_.join(timeout=120)
        assert p_.exitcode == 0
    p0, p1 = torch.tensor(p0), torch.tensor(p1)
    assert torch.equal(p0, p1)
    from gaot_amd.trainer import TrainStep
    dev = torch.device("cuda:0")
    model = _vx_model(seed=300).to(dev).train()
    lat, xs, p, t = _vx_data()
    latd, xd, enc, dec = _vx_graphs(lat, xs, dev)
    ts = TrainStep(model, lr=2e-3, weight_decay=1e-4, use_graph=False)
    b = _VX_BATCHES[0]
    ts.bind(p[b].to(dev), t[b].to(dev), latent_tokens_coord=latd, xcoord=xd[b], encoder_nbrs=[enc[i] for i in b], decoder_nbrs=[dec[i] for i in b])
    for b in _VX_BATCHES:
        ts.step(p[b].to(dev), t[b].to(dev), xcoord=xd[b], encoder_nbrs=[enc[i] for i in b], decoder_nbrs=[dec[i] for i in b])
    ref = _flat(model).cpu()
    assert float((p0 - ref).abs().max()) < 2e-5, float((p0 - ref).abs().max())


def _bench_line_two_ranks(env_extra):
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env={**os.environ, **env_extra})
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_bench_two_ranks_on_one_gpu_over_gloo_line():
    """bench.py's N = 2 code path as the driver launches it (torch.distributed.run, two ranks), with both ranks on cuda:0 over gloo
    (bench.py's test hooks; RCCL needs one device per rank): staged backward, per-phase exchange, MAX-over-ranks timing, `comm`"""
    line = _bench_line_two_ranks({"GAOT_BENCH_FORCE_DEVICE": "0", "GAOT_BENCH_BACKEND": "gloo"})
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 16 and line["value"] > 0
    # four backward phases in two stage groups: [decoder + processor] and [encoder], one all-reduce slice each
    assert line["config"]["staged_backward_phases"] == 4 and line["config"]["stage_groups"] == [[0, 1, 2], [3]]
    assert len(line["comm"]["slices"]) == 2
    assert sum(s["bytes"] for s in line["comm"]["slices"]) >= 4 * line["config"]["params"]
    assert line["comm"]["ms_per_step_without_exchange"] > 0


def test_bench_plain_command_self_launches_two_ranks():
    """`python bench.py --gpus 2` WITHOUT any launcher (WORLD_SIZE unset): bench.py starts the two ranks itself under
    torch.distributed.run and prints exactly one JSON line, from rank 0, whose `comm` proves that two ranks took part"""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"GAOT_BENCH_FORCE_DEVICE": "0", "GAOT_BENCH_BACKEND": "gloo"})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"],
                       capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 16 and line["value"] > 0
    assert line["comm"]["n_ranks_seen"] == 2 and line["comm"]["param_checksums_identical"] is True


def test_bench_plain_command_eight_ranks_on_one_gpu():
    """the driver's 8-GPU command shape on ONE GPU (eight processes share cuda:0, gloo for the exchange): `python bench.py --gpus 8` must
    come back with one JSON line of eight ranks, identical parameters on every rank after the timed steps, the per-slice exchange and the
    exposed communication printed -- so the first run on a real 8-GPU node cannot fail on plumbing (rendezvous, rank -> device, the staged
    backward's two stage groups, max-over-ranks timing, whole-job throughput)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"GAOT_BENCH_FORCE_DEVICE": "0", "GAOT_BENCH_BACKEND": "gloo"})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "2"],
                       capture_output=True, text=True, timeout=1500, cwd=root, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["config"]["global_batch"] == 64 and line["value"] > 0
    assert line["comm"]["n_ranks_seen"] == 8 and line["comm"]["param_checksums_identical"] is True
    assert "exposed_ms_per_step" in line["comm"] and len(line["comm"]["slices"]) == len(line["config"]["stage_groups"]) == 2
    assert abs(line["value"] - 64 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_bench_two_gpus_rccl_line():
    """bench.py exactly as the driver launches it for N = 2: one JSON line with n_gpus 2, weak scaling, and the `comm` object
    (per-slice all-reduce times, exposed communication)"""
    line = _bench_line_two_ranks({})
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["global_batch"] == 16 and line["value"] > 0
    assert line["config"]["staged_backward_phases"] == 4 and len(line["comm"]["slices"]) == len(line["config"]["stage_groups"]) == 2
    assert sum(s["bytes"] for s in line["comm"]["slices"]) >= 4 * line["config"]["params"]


def test_staged_single_rank_equals_unstaged():
    """cut points + per-group graphs without any process group: the same weights as the single-graph step -- bit for bit between the
    staged step's eager and captured forms and between the schedules with the same weight-gradient launches (one group per phase
    pair / per phase), and to fp32 rounding against the unstaged step (the grouped weight-gradient launch picks its K slabs by the
    number of products it holds, so its summation order differs between one launch for the whole pass and one per stage group)"""
    from gaot_amd.trainer import TrainStep
    dev = torch.device("cuda:0")
    lat, x, p, t = _data()
    res = []
    for staged, graph, groups in ((False, True, None), (True, True, None), (True, False, None), (True, True, "each")):
        model = _build(seed=5).to(dev).train()
        ts = TrainStep(model, lr=2e-3, weight_decay=1e-4, use_graph=graph, staged=staged, stage_groups=groups)
        assert ts.staged == staged
        if staged:
            assert ts.stage_groups == ([[k] for k in range(ts.bucket.n_phases)] if groups == "each" else [list(range(ts.bucket.n_phases - 1)), [ts.bucket.n_phases - 1]])
        ts.bind(p.to(dev), t.to(dev), latent_tokens_coord=lat.to(dev), xcoord=x.to(dev))
        for _ in range(3):
            ts.step()
        torch.cuda.synchronize()
        res.append(_flat(model).cpu())
    assert torch.equal(res[1], res[2])                                   # captured == eager, same schedule
    for r in (res[1], res[3]):                                           # either schedule == the unstaged step to rounding
        assert float((r - res[0]).norm() / res[0].norm()) < 2e-6


@pytest.mark.parametrize("graph", [True, False])
def test_whole_staged_trainstep_on_rccl_one_rank_group(graph):
    """the WHOLE staged TrainStep as every rank of an N-GPU job runs it, on RCCL: stage-group hipGraph replays interleaved with
    all_reduce(async_op=True, ReduceOp.AVG) of each group's slice on RCCL's own stream, wait(), optimizer graph -- with a one-rank
    `nccl` group (all a one-GPU box can host) and the exchange forced on (TrainStep.force_comm).  The mean over one rank is the
    identity, so twelve steps must leave the weights BIT-IDENTICAL to the same staged step without any communication: any missing
    stream dependency between a replay and its all-reduce, or an all-reduce reading a slice a later replay is still writing, shows."""
    from gaot_amd.trainer import TrainStep
    port = _free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda:0")
    lat, x, p, t = _data()
    res = {}
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        for comm in (True, False):
            model = _build(seed=5).to(dev).train()
            ts = TrainStep(model, lr=2e-3, weight_decay=1e-4, use_graph=graph, staged=True)
            assert ts.staged and len(ts.stage_groups) == 2
            ts.force_comm, ts.comm_enabled = comm, comm
            ts.bind(p.to(dev), t.to(dev), latent_tokens_coord=lat.to(dev), xcoord=x.to(dev))
            losses = [float(ts.step()) for _ in range(12)]
            torch.cuda.synchronize()
            res[comm] = (_flat(model).cpu(), losses)
    finally:
        dist.destroy_process_group()
    assert torch.equal(res[True][0], res[False][0]) and res[True][1] == res[False][1]
    assert res[True][1][-1] < res[True][1][0]                    # and it trains


def test_rccl_avg_all_reduce_on_flat_bucket_one_rank():
    """the `nccl` (= RCCL) branch of FlatGradBucket.all_reduce_mean: ReduceOp.AVG on the flat buffer and on a phase slice,
    synchronous and async_op, with a one-rank group (all this box can host)."""
    from gaot_amd.trainer import FlatGradBucket
    port = _free_port()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        lin = [torch.nn.Linear(8, 8, bias=False).to(dev) for _ in range(3)]
        b = FlatGradBucket([l.weight for l in lin], phases=[[lin[2].weight], [lin[0].weight, lin[1].weight]])
        b.flat.copy_(torch.arange(b.numel, dtype=torch.float32))
        want = b.flat.clone()
        b.all_reduce_mean(None, _force=True)
        w = b.all_reduce_mean(None, phase=0, async_op=True, _force=True)
        assert w is not None
        w.wait()
        b.all_reduce_mean(None, phase=1, _force=True)
        torch.cuda.synchronize()
        assert torch.equal(b.flat, want)          # the mean over one rank is the identity
    finally:
        dist.destroy_process_group()
