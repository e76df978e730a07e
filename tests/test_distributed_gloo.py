"""CPU, world_size 2 over gloo: the flat-bucket gradient all-reduce and parameter broadcast the trainer uses on RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from a3vt_amd import distributed as adist
    r, w, _ = adist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(rank)                       # different weights per rank until the broadcast
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    adist.broadcast_parameters(net)
    bucket = adist.FlatGradBucket(net.parameters())
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(8, 5, generator=g)            # the global batch; each rank takes its shard
    y = torch.randn(8, 3, generator=g)
    lo, hi = adist.shard_range(8, rank, world)
    for _ in range(3):
        bucket.zero()
        loss = ((net(x[lo:hi]) - y[lo:hi]) ** 2).mean()
        loss.backward()
        bucket.all_reduce_mean()
        opt.step()
    out.put((rank, torch.cat([p.detach().reshape(-1) for p in net.parameters()]).tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_equals_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {k: torch.tensor(v) for k, v in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(res[0], res[1])            # ranks stay in lock-step
    # single-process reference: same init (rank 0's seed), full batch, mean of shard means == full mean (equal shards)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(8, 5, generator=g)
    y = torch.randn(8, 3, generator=g)
    for _ in range(3):
        opt.zero_grad()
        ((net(x) - y) ** 2).mean().backward()
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(res[0], ref, atol=1e-6)


class _Wide(torch.nn.Module):
    """>= 100 parameter tensors, one of them never used in the graph (its gradient stays None after backward)."""

    def __init__(self):
        super().__init__()
        self.layers = torch.nn.ModuleList([torch.nn.Linear(6, 6) for _ in range(60)])   # 120 tensors
        self.unused = torch.nn.Parameter(torch.ones(5))

    def forward(self, x):
        for l in self.layers:
            x = torch.tanh(l(x)) + x
        return x


def _worker_plan(rank, world, port, out):
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import numpy as np
    from a3vt_amd import distributed as adist
    adist.init_from_env("gloo")
    torch.manual_seed(100 + rank)
    net = _Wide()
    adist.broadcast_parameters(net)
    s = adist.seed_rank(0, rank)                  # what Engine.setup does right after the broadcast
    draw = (torch.rand(1).item(), float(np.random.rand()))
    bucket = adist.FlatGradBucket(net.parameters())
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(0)
    data = torch.randn(10, 6, generator=g)        # a 10-item dataset, batch size 2 per rank
    sampler = adist.ShardSampler(10, rank, world, seed=0, shuffle=True, pad=True)
    seen = []
    for epoch in range(2):
        sampler.set_epoch(epoch)
        idx = list(sampler)
        seen.append(idx)
        for k in range(0, len(idx), 2):
            bucket.zero()
            net(data[idx[k:k + 2]]).square().mean().backward()
            assert net.unused.grad is None
            bucket.all_reduce_mean()
            assert net.unused.grad is not None and net.unused.grad.abs().max().item() == 0.0
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
            opt.step()
    # sharded validation: per-rank sums over disjoint shards, summed across ranks
    vs = adist.ShardSampler(7, rank, world, seed=0, shuffle=False, pad=False)
    tot, cnt = torch.zeros(()), torch.zeros(())
    for i in vs:
        tot += float(i + 1)
        cnt += 1
    adist.all_reduce_sum_(tot, cnt)
    out.put((rank, dict(seed=s, draw=draw, seen=seen, params=torch.cat([p.detach().reshape(-1) for p in net.parameters()]).tolist(),
                        val=(tot.item(), cnt.item()), vshard=list(vs))))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_plan_is_disjoint_covering_and_deterministic():
    import numpy as np
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from a3vt_amd import distributed as adist
    for n in (0, 1, 7, 10, 64, 1001):
        for world in (1, 2, 3, 8):
            plans = [adist.rank_plan(r, world, 3, n, epoch=1) for r in range(world)]
            assert len({len(p) for p in plans}) == 1                                   # same number of steps everywhere
            cat = np.concatenate([p.indices for p in plans])
            assert set(cat.tolist()) == set(range(n)) and len(cat) - n < world          # covering, < world wrapped extras
            assert [p.seed for p in plans] == [3 + r for r in range(world)]             # seed + rank
            again = [adist.rank_plan(r, world, 3, n, epoch=1) for r in range(world)]
            assert all(np.array_equal(a.indices, b.indices) for a, b in zip(plans, again))
            exact = [adist.rank_plan(r, world, 3, n, epoch=1, pad=False).indices for r in range(world)]
            assert sorted(np.concatenate(exact).tolist()) == list(range(n))             # validation: exactly disjoint
            if n > 10:
                other = adist.rank_plan(0, world, 3, n, epoch=2).indices
                assert not np.array_equal(other, plans[0].indices)                      # reshuffled per epoch
    order = adist.rank_plan(0, 1, 3, 10, shuffle=False).indices
    assert order.tolist() == list(range(10))


def test_two_rank_engine_plan_bucket_and_sharded_validation():
    """What ``Engine`` does with WORLD_SIZE=2, on CPU over gloo: broadcast, ``seed + rank`` reseed, per-epoch sharded
    sampler, the flat gradient bucket over 121 tensors (one unused), sharded validation with summed scores."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_plan, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]["seed"] == 0 and res[1]["seed"] == 1 and res[0]["draw"] != res[1]["draw"]   # ranks draw differently
    assert torch.equal(torch.tensor(res[0]["params"]), torch.tensor(res[1]["params"]))        # and stay in lock-step
    for epoch in range(2):
        a, b = res[0]["seen"][epoch], res[1]["seen"][epoch]
        assert len(a) == len(b) == 5 and set(a) | set(b) == set(range(10)) and not set(a) & set(b)
    assert res[0]["seen"][0] != res[0]["seen"][1]
    assert sorted(res[0]["vshard"] + res[1]["vshard"]) == list(range(7))
    assert res[0]["val"] == res[1]["val"] == (28.0, 7.0)


class _ThreeStage(torch.nn.Module):
    """Shape of ``Deformation``: ``first`` feeds stage 1, ``shared`` stages 2 and 3 (so its gradients are final before the
    backward pass has reached stage 1), a BatchNorm branch with running statistics (the image encoders')."""

    def __init__(self):
        super().__init__()
        self.first = torch.nn.Linear(6, 6)
        self.shared = torch.nn.Linear(6, 6)
        self.bn = torch.nn.BatchNorm1d(6)

    def forward(self, x):
        x = torch.tanh(self.first(self.bn(x))) + x
        x = torch.tanh(self.shared(x)) + x
        return torch.tanh(self.shared(x)) + x


def _worker_two_chunks(rank, world, port, out):
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from a3vt_amd import distributed as adist
    adist.init_from_env("gloo")
    torch.manual_seed(50 + rank)
    net = _ThreeStage()
    adist.broadcast_parameters(net)
    adist.broadcast_buffers(net)
    bucket = adist.FlatGradBucket(net.parameters(), early=list(net.shared.parameters()))
    assert bucket.n_early == 2 and bucket.params[0] is net.shared.weight and bucket.early_numel == 42
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(16, 6, generator=g)
    lo, hi = adist.shard_range(16, rank, world)
    started_early = []
    first_grad_seen = []
    net.first.weight.register_post_accumulate_grad_hook(lambda p: first_grad_seen.append(bucket._early_done))
    for _ in range(3):
        bucket.zero()
        net(x[lo:hi]).square().mean().backward()
        # the early chunk's reduce was started from inside the backward pass, before stage 1's gradient existed
        started_early.append(bucket._early_done and bucket._early_work is not None)
        bucket.all_reduce_mean()
        assert bucket._early_work is None
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
        opt.step()
    stats_before = net.bn.running_mean.clone()     # per-rank batch statistics: the ranks have drifted apart
    adist.broadcast_buffers(net)                   # what Engine.validate does first
    out.put((rank, dict(params=torch.cat([p.detach().reshape(-1) for p in net.parameters()]).tolist(),
                        started_early=started_early, first_grad_seen=first_grad_seen,
                        stats_before=stats_before.tolist(), stats_after=net.bn.running_mean.tolist(),
                        tracked=int(net.bn.num_batches_tracked))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_chunk_async_reduce_and_batchnorm_buffers():
    """The gradient exchange in two chunks (the early one started asynchronously by the post-accumulate hooks, inside the
    backward pass) gives the single-process result; BatchNorm running statistics differ per rank after training and are
    rank 0's everywhere after ``broadcast_buffers``."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_two_chunks, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(torch.tensor(res[0]["params"]), torch.tensor(res[1]["params"]))
    for r in range(world):
        assert res[r]["started_early"] == [True] * 3
        assert res[r]["first_grad_seen"] == [True] * 3          # stage 1's gradient arrived after the early reduce had started
    assert res[0]["stats_before"] != res[1]["stats_before"]      # no SyncBN, as the reference
    assert res[0]["stats_after"] == res[1]["stats_after"] == res[0]["stats_before"]
    assert res[0]["tracked"] == res[1]["tracked"] == 3
    # single-process reference on the two half batches averaged = data-parallel step (BatchNorm normalises per shard)
    torch.manual_seed(50)
    net = _ThreeStage()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(16, 6, generator=g)
    for _ in range(3):
        opt.zero_grad()
        (0.5 * (net(x[:8]).square().mean() + net(x[8:]).square().mean())).backward()
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    assert torch.allclose(torch.tensor(res[0]["params"]), ref, atol=2e-6)


def _worker_engine_validation(rank, world, port, out):
    """Engine.validate / check_values with WORLD_SIZE = 2 and INJECTED loaders (no sharded sampler): the ranks score the
    same batches with different surface samples (per-rank seeds), so their local scores differ; the decisions must not."""
    import sys
    import types
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from a3vt_amd import distributed as adist
    from a3vt_amd.pterotactyl.reconstruction.vision import train as vtrain
    adist.init_from_env("gloo")
    eng = object.__new__(vtrain.Engine)            # no mesh / HIP library on this box: only the control flow is under test
    eng.rank, eng.world, eng.local_rank = rank, world, rank
    eng.args = types.SimpleNamespace(eval=False, loss_coeff=1.0, number_points=8, exp_id="t", patience=2)
    eng.initial_mesh = torch.zeros(4, 3)
    eng.mesh_info = {"faces_i32": None}
    eng.encoder = torch.nn.Linear(1, 1)
    eng.encoder.forward = lambda img, charts: (charts, None)
    eng.epoch, eng.best_loss, eng.last_improvement = 0, 10000, 0
    saves = []
    eng.save = lambda: saves.append(eng.epoch)
    torch.manual_seed(1000 + rank)                 # per-rank sample stream (Engine.setup seeds seed + rank)
    scores = iter([5.0, 4.0, 4.5, 4.6, 4.7])       # epoch scores before the per-rank sampling noise
    base = {"v": 0.0}
    vtrain.model.prepare_mesh = lambda batch, mesh, args: batch["img"]
    vtrain.utils.chamfer_distance = lambda verts, faces, gt, num: base["v"] + 0.2 * torch.rand(verts.shape[0])
    vtrain.data_loaders.DevicePrefetcher = lambda loader, dev: loader

    class W:
        def add_scalars(self, *a, **k):
            pass
    loader = [{"img": torch.zeros(3, 1), "gt_points": torch.zeros(3, 8, 3)} for _ in range(2)]   # injected: no sampler
    history, stopped = [], None
    for epoch in range(5):
        eng.epoch = epoch
        base["v"] = next(scores)
        eng.validate(loader, W())
        history.append(eng.current_loss)
        try:
            eng.check_values()
        except StopIteration:
            stopped = epoch
            break
    out.put((rank, dict(history=history, stopped=stopped, saves=saves, best=eng.best_loss)))
    dist.barrier()
    dist.destroy_process_group()


def test_validation_score_and_early_stop_agree_across_ranks_with_injected_loaders():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_engine_validation, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]["history"] == res[1]["history"]                     # the same score on every rank, every epoch
    assert res[0]["stopped"] == res[1]["stopped"] == 3                # 5.x, 4.x (best), 4.5+, 4.6+ -> patience 2 spent
    assert res[0]["best"] == res[1]["best"]
    assert res[0]["saves"] == [0, 1] and res[1]["saves"] == [0, 1]    # (save() itself writes on rank 0 only)


def _worker_late(rank, world, port, out):
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from a3vt_amd import distributed as adist
    adist.init_from_env("gloo")
    torch.manual_seed(3)
    a, b, c = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 2)
    params = list(a.parameters()) + list(b.parameters()) + list(c.parameters())
    bucket = adist.FlatGradBucket(params, early=list(a.parameters()) + list(b.parameters()))
    x = torch.randn(6, 4, generator=torch.Generator().manual_seed(rank))
    events = []
    for step in range(3):
        bucket.zero()
        # rank 1 leaves `b` out of the graph in step 0 (its countdown then expects two early gradients, not four) and uses it
        # again in step 1: the countdown fires after a's two, b's arrive LATE — on rank 1 only
        use_b = not (rank == 1 and step == 0)
        # c(a(x)) first in the expression: autograd reaches b's parameters BEFORE a's in the backward pass?  No order is assumed:
        # what matters is that MORE early parameters receive gradients than the countdown expects
        y = c(a(x)).sum() + (b(x).sum() if use_b else 0.0)
        y.backward()
        try:
            bucket.all_reduce_mean()
            events.append("ok")
        except RuntimeError as e:
            events.append("late-here" if "on this rank" in str(e) else "late-peer")
            break
    out.put((rank, events))
    dist.barrier()
    dist.destroy_process_group()


def test_late_early_gradient_raises_on_every_rank():
    """a rank whose early chunk was gathered before its last early gradient arrived used to raise ALONE, after both
    collectives: its peers applied a step whose early average missed that gradient and met mis-paired collectives afterwards.
    The late flag now travels with the [rest] reduce: with two ranks over gloo, rank 1 is late in step 1 and BOTH ranks raise
    in that step, after the same two collectives (nobody hangs, nobody steps)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_late, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] == ["ok", "late-here"], res
    assert res[0] == ["ok", "late-peer"], res
