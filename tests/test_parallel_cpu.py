"""Data-parallel host logic on CPU: world_size-2 gloo processes, one flat all-reduce per step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eagcn_amd.parallel import GradientAllReducer, shard_range


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 256, 4097):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)                                   # identical replicas
    model = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.ReLU(), torch.nn.Linear(4, 3))
    unused = torch.nn.Parameter(torch.zeros(2))            # a parameter that never gets a gradient
    params = list(model.parameters()) + [unused]
    g = torch.Generator().manual_seed(100)
    x_all = torch.randn(8, 5, generator=g)
    lo, hi = shard_range(8, rank, world)
    model(x_all[lo:hi]).pow(2).sum().backward()            # local shard, sum-reduced loss
    GradientAllReducer(params)()
    flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    # oracle: the same model on the global batch; averaged shard grads == global grad / world
    ref = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.ReLU(), torch.nn.Linear(4, 3))
    ref.load_state_dict(model.state_dict())
    ref(x_all).pow(2).sum().backward()
    want = torch.cat([p.grad.reshape(-1) for p in ref.parameters()]) / world
    q.put((rank, float((flat - want).abs().max()), unused.grad is None))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gradient_allreduce_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, err, unused_none in res:
        assert err < 1e-6, (rank, err)
        assert unused_none
