"""Data-parallel step on N ranks (one process per GPU, RCCL), run by tests/test_gpu_dist.py -- under
``python -m torch.distributed.run --nproc-per-node N`` when the box has N >= 2 GPUs, or stand-alone with EAGCN_FORCE_DIST=1 /
WORLD_SIZE=1 (same code path, one rank).

What every rank does: its contiguous shard of ONE global synthetic batch (all shards padded to the global N), the graph-mode
fused training step with the gradient average captured INSIDE the step graph (GraphRunner._call_backward_comm: upper bucket
started before the first layer's backward) and the global BCE normalisation ('dp' scale, a 1-element collective issued with the
batch preparation).  What is checked, on every rank, against the CPU oracle (oracle/eagcn_ref.py):
  * local-BN (default): averaged gradients == sum over shards of the gradient of S_r / n_total, every shard an oracle run of
    its own (BatchNorm over the shard's rows) -- SURVEY.md 8e "BN modes (i)";
  * sync_bn=True: averaged gradients, loss and outputs == ONE oracle run on the concatenated batch -- "BN modes (ii)";
  * all ranks hold bit-identical averaged gradients; the runner reports that the collective really was captured in the graph.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

from eagcn_amd import EAGCN
from eagcn_amd.parallel import GradientAllReducer, init_distributed, shard_range
from eagcn_amd.synthetic import bce_weights, make_batch
from oracle.eagcn_ref import RefEAGCN, weights_init_      # checker only

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


def bce_sum(out, l):
    w = torch.tensor(bw, dtype=out.dtype)
    wt = ((l == 1).to(out.dtype) * w[:, 0].view(1, -1) + (l == 0).to(out.dtype) * w[:, 1].view(1, -1)).view(-1)
    return torch.nn.functional.binary_cross_entropy_with_logits(out.view(-1), l.to(out.dtype).view(-1), weight=wt, reduction='sum')


n_tot = float(((labels_all == 1) | (labels_all == 0)).sum())


def oracle(sync):
    """(loss of the global batch, gradients) as the reference computes them: one run on the concatenated batch (sync) or one
    run per shard with its own BatchNorm statistics (local)."""
    m = RefEAGCN(9, 24, W1, W2, 32, 16, T, 0.0, n_layers=2).double()
    m.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
    m.train()
    pieces = [(0, BS * world)] if sync else [shard_range(BS * world, r, world) for r in range(world)]
    total = 0.0
    for a, b in pieces:
        d = [t[a:b].double() if t.is_floating_point() else t[a:b] for t in dense_all]
        out, _, _ = m(*d)
        loss = bce_sum(out, labels_all[a:b]) / n_tot
        loss.backward()
        total += float(loss)
    return total, {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}


# (local-BN twice: the second time with the batch preparation -- and with it the one-element 'dp' loss-scale collective -- on the
#  side stream, i.e. an eager collective issued while the previous step's graph with its captured all-reduces may still be replaying
#  on the same communicator: the configuration bench.py and examples/train_synth.py run)
for sync, overlap in ((False, False), (False, True), (True, False)):
    torch.manual_seed(1)
    model = EAGCN(9, 24, *W1, *W2, 32, 16, T, 0.0, n_layers=2, graph=True, validate='deferred', sync_bn=sync,
                  overlap_index=overlap).to(dev).train()
    model.load_state_dict(ref.state_dict(), strict=True)
    red = GradientAllReducer(model.parameters(), model=model)
    want_loss, want = oracle(sync)
    nsteps = 12 if overlap else 5
    for step in range(nsteps):                    # eager + capture on both slots, then replays
        for p in model.parameters():
            p.grad = None
        loss, (out, _, _) = model.fused_step(shard, labels, 'class', bw_dev, 'dp', reducer=red)
        if overlap and 4 <= step < nsteps - 1:
            continue                              # replays back to back, no host synchronisation: the next step's eager collective
                                                  # is issued while this step's graph (with its captured all-reduces) is in flight
        torch.cuda.synchronize()
        # the ranks' losses c_r * S_r / n_r average to the global loss
        lt = loss.detach().clone().reshape(1)
        dist.all_reduce(lt, op=dist.ReduceOp.SUM)
        got_loss = float(lt) / world
        assert abs(got_loss - want_loss) <= 2e-5 * max(1.0, abs(want_loss)), (sync, step, got_loss, want_loss)
        scale = max(v.abs().max().item() for v in want.values())
        worst = (0.0, '')
        for k, p in model.named_parameters():
            if k not in want:
                continue
            g = p.grad.detach().double().cpu()
            e = (g - want[k]).abs().max().item()
            tol = 2e-5 * want[k].abs().max().item() + 2e-6 * scale
            assert e <= tol, ('sync' if sync else 'local', step, k, e, tol)
            if want[k].abs().max().item() > 1e-6 * scale:          # (analytically-zero gradients have no own scale)
                worst = max(worst, (e / want[k].abs().max().item(), k))
        # identical on every rank
        flat = model.flat_grad_buffer()
        assert flat is not None
        chk = flat.double().sum().reshape(1)
        gathered = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(gathered, chk)
        assert all(torch.equal(gathered[0], c) for c in gathered), [float(c) for c in gathered]
    runner = next(iter(model._runners.values()))
    print('rank %d %s-BN%s: loss %.6f (oracle %.6f), worst gradient error / own max %.1e (%s), all-reduce %s, sync-BN hook calls %s'
          % (rank, 'sync' if sync else 'local', ' (side-stream batch preparation)' if overlap else '', got_loss, want_loss, worst[0], worst[1],
             'captured in the step graph' if runner.comm_in_graph else 'host-issued after the graph (capture of the collective failed)',
             model.plan().stats.calls if sync else '-'), flush=True)
    if sync:
        assert model.plan().stats.calls > 0 and model.plan().stats.error is None, model.plan().stats.error
        # ONE collective per layer and direction (the sums of all K views travel as one vector of 2 * width + 1 doubles:
        # csrc/layer.hip eagcn_layer_forward / _backward) + one per head BatchNorm and direction; the hook runs only while a step
        # is issued eagerly or captured (2 slots x (eager + capture)), replays carry the collectives as graph nodes
        assert model.plan().stats.calls <= 4 * (2 * 2 + 6), model.plan().stats.calls
    else:
        torch.cuda.synchronize()
        model.release_graphs()            # the local-BN model's step graphs (RCCL nodes) go before the next model captures its own
# ---- the whole data-parallel training iteration: in-graph gradient average, then the parameter update of FlatAdam as the last
# launch of the same graph (it must see the AVERAGED gradients): three steps against torch.optim.Adam on the float64 oracle that
# steps on the gradient of the global loss (local-BN: the sum over the shards' own runs)
from eagcn_amd.optim import FlatAdam
from eagcn_amd import training
torch.cuda.synchronize()
model.release_graphs()                    # (the sync-BN model's step graphs)
torch.manual_seed(1)
model = EAGCN(9, 24, *W1, *W2, 32, 16, T, 0.0, n_layers=2, graph=True, validate='deferred').to(dev).train()
model.load_state_dict(ref.state_dict(), strict=True)
opt = FlatAdam(model, lr=1e-3, weight_decay=1e-4)
red = GradientAllReducer(model.parameters(), model=model)
m64 = RefEAGCN(9, 24, W1, W2, 32, 16, T, 0.0, n_layers=2).double()
m64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
m64.train()
o64 = torch.optim.Adam(m64.parameters(), lr=1e-3, weight_decay=1e-4)
for step in range(3):
    training.train_step(model, opt, shard, labels, 'class', bw_dev, dp_global_norm=True, reducer=red)
    o64.zero_grad()
    for a, b in [shard_range(BS * world, r, world) for r in range(world)]:
        d = [t[a:b].double() if t.is_floating_point() else t[a:b] for t in dense_all]
        (bce_sum(m64(*d)[0], labels_all[a:b]) / n_tot).backward()
    o64.step()
torch.cuda.synchronize()
worst = (0.0, '')
sd64 = dict(m64.named_parameters())
for k, p in model.named_parameters():
    if sd64[k].grad is None or k.endswith('graph_conv.bias') or k == 'Graph_BN.bias':
        continue                                   # (never trained / analytically-zero gradients: Adam turns their rounding noise into +-lr steps)
    e = (p.detach().double().cpu() - sd64[k].detach()).abs().max().item() / max(sd64[k].detach().abs().max().item(), 1e-3)
    worst = max(worst, (e, k))
print('rank %d: three data-parallel FlatAdam iterations (update inside the step graph, behind the captured all-reduce): worst parameter '
      'distance to torch.optim.Adam on the float64 oracle %.1e (%s)' % (rank, worst[0], worst[1]), flush=True)
assert worst[0] < 2e-4, worst                      # three Adam steps of lr 1e-3 move a parameter by up to 3e-3: a missed average or a
                                                   # stale gradient shows as 1e-3, rounding noise in small second moments as 1e-5
chk = opt.flat.double().sum().reshape(1)
gathered = [torch.zeros_like(chk) for _ in range(world)]
dist.all_gather(gathered, chk)
assert all(torch.equal(gathered[0], c) for c in gathered), 'the replicas diverged'
torch.cuda.synchronize()
model.release_graphs()
# Orderly teardown, in dependency order: (1) every pending collective has completed on every rank, (2) the captured step graphs
# -- they hold RCCL kernel nodes -- and the static buffers are destroyed explicitly (EAGCN.release_graphs), not whenever the
# garbage collector gets to them, (3) the process group (communicator + its watchdog thread) goes, (4) the process leaves
# through os._exit.  The SIGABRT this script used to die of once in ~25 runs came from a background thread of the process
# group AFTER all checks had passed, while the interpreter was tearing down module globals and static destructors of the HIP /
# RCCL runtimes were already running in the main thread; with nothing of ours alive and no interpreter finalisation that
# window does not exist.  tests/test_gpu_dist.py no longer tolerates a rank that ends with a signal.
dist.barrier()
torch.cuda.synchronize()
model.release_graphs()
del model, red, runner
import gc
gc.collect()
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print('DIST_MULTI_OK rank %d of %d' % (rank, world), flush=True)
sys.stdout.flush()
sys.stderr.flush()
os._exit(0)
