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
