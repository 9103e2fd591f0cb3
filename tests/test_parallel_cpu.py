"""Data-parallel host logic on CPU: world_size-2 gloo processes, one flat all-reduce per step."""
import os
import sys
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


# ---- the flat-buffer path (what graph mode / grad_mode='direct' use), BCE global normalisation, sync-BN statistics -----
class _FlatModel(torch.nn.Module):
    """Stand-in for EAGCN in grad_mode='direct': every gradient is a view of ONE flat buffer."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Parameter(torch.randn(4, 3))
        self.b = torch.nn.Parameter(torch.randn(5))
        self.unused = torch.nn.Parameter(torch.zeros(2))
        self._flat = None

    def flat_grad_buffer(self):
        return self._flat

    def backward_into_flat(self, ga, gb):
        self._flat = torch.cat([ga.reshape(-1), gb.reshape(-1)]).clone()
        self.a.grad = self._flat[:12].view(4, 3)
        self.b.grad = self._flat[12:]


def _worker_flat(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from eagcn_amd.parallel import dp_loss_scale, sync_bn_backward_stats, sync_bn_forward_stats
    from oracle.eagcn_ref import classification_loss
    torch.manual_seed(0)
    model = _FlatModel()
    g = torch.Generator().manual_seed(7)
    ga_all, gb_all = torch.randn(world, 4, 3, generator=g), torch.randn(world, 5, generator=g)
    model.backward_into_flat(ga_all[rank], gb_all[rank])
    red = GradientAllReducer(model.parameters(), model=model)
    flat_before = model.flat_grad_buffer().data_ptr()
    red()
    ok_inplace = model.flat_grad_buffer().data_ptr() == flat_before and model.a.grad.data_ptr() == flat_before
    err_flat = max(float((model.a.grad - ga_all.mean(0)).abs().max()), float((model.b.grad - gb_all.mean(0)).abs().max()))
    # the bucketed form the graph-mode step uses (GraphRunner._call_backward_comm): upper bucket started asynchronously, work done
    # in between, lower bucket afterwards -- same averages, in place
    model.backward_into_flat(ga_all[rank], gb_all[rank])
    flat = model.flat_grad_buffer()
    assert red.active()
    upper = red.start(flat[12:])
    filler = torch.randn(64, 64) @ torch.randn(64, 64)              # (stands for the first layer's backward)
    upper.wait()
    red.start(flat[:12]).wait()
    err_flat = max(err_flat, float((model.a.grad - ga_all.mean(0)).abs().max()), float((model.b.grad - gb_all.mean(0)).abs().max()),
                   0.0 * float(filler.sum()))

    # ---- BCE: per-shard normalisation vs the reference's normalisation on the concatenated batch (train.py:328-331)
    T = 3
    logits_all = torch.randn(10, T, generator=g)
    labels_all = torch.tensor([[1, 0, -1], [0, -1, -1], [-1, -1, -1], [1, 1, 0], [0, 0, 0],      # rank 0: 8 labelled
                               [1, -1, -1], [-1, -1, -1], [-1, 0, -1], [-1, -1, -1], [-1, -1, 1]], dtype=torch.float32)
    bw = [[5.0, 1.2], [3.0, 1.1], [7.0, 1.05]]
    w = torch.nn.Parameter(torch.randn(T, T, generator=g))
    lo, hi = shard_range(10, rank, world)
    out = logits_all[lo:hi] @ w
    scale = dp_loss_scale(labels_all[lo:hi])
    (classification_loss(out, labels_all[lo:hi], bw) * scale).backward()
    GradientAllReducer([w])()
    w_ref = torch.nn.Parameter(w.detach().clone())
    classification_loss(logits_all @ w_ref, labels_all, bw).backward()
    err_bce = float((w.grad - w_ref.grad).abs().max() / w_ref.grad.abs().max())

    # ---- sync-BN: statistics and backward means of the concatenated batch from per-rank partial sums
    Fc, eps = 6, 1e-5
    rows = [7, 12]                                           # unequal shards (every shard padded to its own N_pad)
    x_all = torch.randn(sum(rows), Fc, generator=g, dtype=torch.float64) * 2.0 + 0.5
    gcot = torch.randn(sum(rows), Fc, generator=g, dtype=torch.float64)
    gamma = torch.rand(Fc, generator=g, dtype=torch.float64) + 0.5
    r0 = sum(rows[:rank])
    x, dh = x_all[r0:r0 + rows[rank]], gcot[r0:r0 + rows[rank]]
    mean, var, unbiased, M = sync_bn_forward_stats(x.sum(0), (x * x).sum(0), rows[rank])
    inv = 1.0 / torch.sqrt(var + eps)
    xhat = (x - mean) * inv
    s1, s2 = sync_bn_backward_stats(dh.sum(0), (dh * xhat).sum(0))
    dx = gamma * inv * (dh - s1 / M - xhat * (s2 / M))
    dgamma_local = (dh * xhat).sum(0)
    dgam = dgamma_local.clone()
    dist.all_reduce(dgam)
    xr = x_all.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    rm, rv = torch.zeros(Fc, dtype=torch.float64), torch.ones(Fc, dtype=torch.float64)
    y = torch.nn.functional.batch_norm(xr, rm, rv, gr, torch.zeros(Fc, dtype=torch.float64), True, 1.0, eps)
    (y * gcot).sum().backward()
    err_bn = max(float((dx - xr.grad[r0:r0 + rows[rank]]).abs().max()), float((dgam - gr.grad).abs().max()),
                 float((mean - rm).abs().max()), float((unbiased - rv).abs().max()),
                 float(((x - mean) * inv * gamma - y.detach()[r0:r0 + rows[rank]]).abs().max()))
    q.put((rank, ok_inplace, err_flat, err_bce, err_bn, model.unused.grad is None))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_buffer_allreduce_bce_global_norm_and_sync_bn_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_flat, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, ok_inplace, err_flat, err_bce, err_bn, unused_none in res:
        assert ok_inplace, 'the flat gradient buffer was not reduced in place (rank %d)' % rank
        assert err_flat < 1e-6, (rank, err_flat)
        assert err_bce < 1e-5, (rank, err_bce)
        assert err_bn < 1e-9, (rank, err_bn)
        assert unused_none


# ---- four ranks, uneven shards (10 molecules -> 3 / 3 / 2 / 2), the BUCKETED asynchronous form of the graph-mode step -------------
def _worker_ws4(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from eagcn_amd.parallel import dp_loss_scale
    from oracle.eagcn_ref import classification_loss
    torch.manual_seed(0)                                   # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    g = torch.Generator().manual_seed(5)
    n_items = 10
    x_all = torch.randn(n_items, 6, generator=g)
    labels_all = torch.tensor([[1, 0, -1], [0, -1, -1], [-1, -1, -1], [1, 1, 0], [0, 0, 0], [1, -1, -1], [-1, -1, -1], [-1, 0, -1],
                               [-1, -1, -1], [-1, -1, 1]], dtype=torch.float32)
    bw = [[5.0, 1.2], [3.0, 1.1], [7.0, 1.05]]
    lo, hi = shard_range(n_items, rank, world)
    sizes = [b - a for a, b in (shard_range(n_items, r, world) for r in range(world))]
    scale = dp_loss_scale(labels_all[lo:hi])               # this shard may hold no labelled entry at all (rank 3 has one)
    (classification_loss(net(x_all[lo:hi]), labels_all[lo:hi], bw) * scale).backward()
    params = list(net.parameters())
    # the flat gradient buffer of graph mode: .grad are views of it; layer 1 = first bucket [0, cut), the rest = upper bucket
    flat = torch.cat([p.grad.reshape(-1) for p in params]).clone()
    o = 0
    for p in params:
        p.grad = flat[o:o + p.numel()].view_as(p)
        o += p.numel()
    cut = params[0].numel() + params[1].numel()
    red = GradientAllReducer(params)
    upper = red.start(flat[cut:])                          # started first (the upper layers finish their backward first) ...
    filler = torch.randn(32, 32) @ torch.randn(32, 32)     # ... the first layer's backward runs beside it ...
    upper.wait()
    red.start(flat[:cut]).wait()                           # ... and its own bucket follows
    ref = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    ref.load_state_dict(net.state_dict())
    classification_loss(ref(x_all), labels_all, bw).backward()
    want = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    err = float((flat - want).abs().max() / want.abs().max()) + 0.0 * float(filler.sum())
    views_ok = all(p.grad.data_ptr() >= flat.data_ptr() for p in params)
    q.put((rank, sizes, err, views_ok))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_bucketed_allreduce_uneven_shards_world4_gloo():
    """SURVEY 8(e) at four ranks: 10 molecules shard 3 / 3 / 2 / 2; the global BCE normalisation ('dp' scale) times the bucketed,
    asynchronously started average of the flat gradient buffer == the gradient of the reference's loss on the whole batch."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ws4, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, sizes, err, views_ok in res:
        assert sizes == [3, 3, 2, 2], sizes
        assert err < 1e-5, (rank, err)
        assert views_ok
