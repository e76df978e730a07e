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
