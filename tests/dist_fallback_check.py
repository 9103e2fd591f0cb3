"""Run by tests/test_gpu_dist.py with EAGCN_COMM_IN_GRAPH=0 (world 1 with EAGCN_FORCE_DIST=1, or N ranks under torchrun): the
HOST-ISSUED gradient average behind the step graph -- the fallback a data-parallel run takes when the collective cannot be
captured.  Regression test of the round-4 finding: on a slot's first eager step the .grad views are not attached
yet (zero_grad(set_to_none) left them None), so a reducer that walks p.grad reduced NOTHING and FlatAdam stepped on local,
unaveraged gradients.  Checked: every step issues exactly one average of the whole flat buffer BEFORE the update (counted), and
four FlatAdam iterations track torch.optim.Adam on the float64 oracle stepping on the global gradient on every rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

from eagcn_amd import EAGCN, training
from eagcn_amd.optim import FlatAdam
from eagcn_amd.parallel import GradientAllReducer, init_distributed, shard_range
from eagcn_amd.synthetic import bce_weights, make_batch
from oracle.eagcn_ref import RefEAGCN, weights_init_      # checker only

assert os.environ.get('EAGCN_COMM_IN_GRAPH') == '0'
rank, world, local = init_distributed()
assert dist.is_initialized() and dist.get_backend() == 'nccl'
dev = torch.device('cuda', local)
T, BS = 3, 12
W1, W2 = [12, 8, 8, 8, 8], [20, 12, 12, 12, 12]
torch.manual_seed(0)
ref = RefEAGCN(9, 24, W1, W2, 32, 16, T, 0.0, n_layers=2)
weights_init_(ref)
mb = make_batch(B=BS * world, n_max=30, n_med=10, rel_channels=(9, 4, 2, 2, 2), seed=11, n_tasks=T)
dense_all = mb.dense()
labels_all = torch.from_numpy(mb.labels)
bw = bce_weights(T)
lo, hi = shard_range(BS * world, rank, world)
shard = [t[lo:hi].to(dev) for t in dense_all]
labels = labels_all[lo:hi].to(dev)
bw_dev = torch.tensor(bw, dtype=torch.float32, device=dev)
n_tot = float(((labels_all == 1) | (labels_all == 0)).sum())


def bce_sum(out, l):
    w = torch.tensor(bw, dtype=out.dtype)
    wt = ((l == 1).to(out.dtype) * w[:, 0].view(1, -1) + (l == 0).to(out.dtype) * w[:, 1].view(1, -1)).view(-1)
    return torch.nn.functional.binary_cross_entropy_with_logits(out.view(-1), l.to(out.dtype).view(-1), weight=wt, reduction='sum')


torch.manual_seed(1)
model = EAGCN(9, 24, *W1, *W2, 32, 16, T, 0.0, n_layers=2, graph=True, validate='deferred').to(dev).train()
model.load_state_dict(ref.state_dict(), strict=True)
opt = FlatAdam(model, lr=1e-3, weight_decay=1e-4)
reduced = []                                       # elements averaged per call, in issue order


class CountingReducer(GradientAllReducer):
    def start(self, t):
        reduced.append(t.numel())
        return super().start(t)

    def __call__(self):
        flat = self._model_flat_buffer()
        reduced.append(flat.numel() if flat is not None else 0)
        return super().__call__()


red = CountingReducer(model.parameters(), model=model)
m64 = RefEAGCN(9, 24, W1, W2, 32, 16, T, 0.0, n_layers=2).double()
m64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
m64.train()
o64 = torch.optim.Adam(m64.parameters(), lr=1e-3, weight_decay=1e-4)
n_flat = opt.flat.numel()
for step in range(5):                              # slot 0 eager+capture, slot 1 eager+capture, then replays
    before = len(reduced)
    training.train_step(model, opt, shard, labels, 'class', bw_dev, dp_global_norm=True, reducer=red)
    torch.cuda.synchronize()
    runner = next(iter(model._runners.values()))
    assert runner.comm_in_graph is False
    flat = model.flat_grad_buffer()
    assert flat is not None and flat.numel() == n_flat
    o64.zero_grad()
    for a, b in [shard_range(BS * world, r, world) for r in range(world)]:
        d = [t[a:b].double() if t.is_floating_point() else t[a:b] for t in dense_all]
        (bce_sum(m64(*d)[0], labels_all[a:b]) / n_tot).backward()
    if step == 0:
        # same parameters on both sides: the averaged gradient (mean_r of the gradients of c_r L_r) is the gradient of the global loss
        want = {k: p.grad.clone() for k, p in m64.named_parameters() if p.grad is not None}
        scale = max(v.abs().max().item() for v in want.values())
        for k, p in model.named_parameters():
            if k in want and p.grad is not None:
                e = (p.grad.detach().double().cpu() - want[k]).abs().max().item()
                tol = 3e-5 * want[k].abs().max().item() + 3e-6 * scale
                assert e <= tol, (step, k, e, tol)
    o64.step()
    got = reduced[before:]
    assert sum(got) == n_flat and len(got) == 1, 'step %d: expected ONE average of the whole flat gradient buffer (%d floats), saw %r' % (step, n_flat, got)
torch.cuda.synchronize()
worst = (0.0, '')
sd64 = dict(m64.named_parameters())
for k, p in model.named_parameters():
    if sd64[k].grad is None or k.endswith('graph_conv.bias') or k == 'Graph_BN.bias':
        continue
    e = (p.detach().double().cpu() - sd64[k].detach()).abs().max().item() / max(sd64[k].detach().abs().max().item(), 1e-3)
    worst = max(worst, (e, k))
print('rank %d: five data-parallel FlatAdam iterations with the HOST-ISSUED gradient average: worst parameter distance to the float64 '
      'oracle %.1e (%s)' % (rank, worst[0], worst[1]), flush=True)
assert worst[0] < 3e-4, worst
chk = opt.flat.double().sum().reshape(1)
gathered = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(gathered, chk)
assert all(torch.equal(gathered[0], c) for c in gathered), 'the replicas diverged'
torch.cuda.synchronize()
model.release_graphs()
dist.barrier()
del model, red, runner
import gc
gc.collect()
torch.cuda.synchronize()
dist.destroy_process_group()
print('DIST_FALLBACK_OK rank %d of %d' % (rank, world), flush=True)
sys.stdout.flush()
sys.stderr.flush()
os._exit(0)
