"""Run by tests/test_gpu_dist.py in a subprocess (EAGCN_FORCE_DIST=1, WORLD_SIZE=1): RCCL initialises, the gradient
all-reduce runs in place on the flat gradient buffer of a graph-mode model, the BCE normalisation collective runs, and
none of it changes a single gradient bit pattern beyond fp32 rounding of x/1."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from eagcn_amd import EAGCN, losses
from eagcn_amd.parallel import GradientAllReducer, init_distributed
from eagcn_amd.synthetic import bce_weights, make_batch

rank, world, local = init_distributed()
assert dist.is_initialized() and dist.get_backend() == 'nccl' and world == 1
dev = torch.device('cuda', local)
torch.manual_seed(0)
mb = make_batch(B=16, n_max=40, n_med=12, rel_channels=(9, 4, 2, 2, 2), seed=3, n_tasks=4)
dense = [t.to(dev) for t in mb.dense()]
labels = torch.from_numpy(mb.labels).to(dev)
w = torch.tensor(bce_weights(4), device=dev)
res = {}
for mode in ('plain', 'dist'):
    torch.manual_seed(1)
    model = EAGCN(9, 24, *[12, 8, 8, 8, 8], *[20, 10, 10, 10, 10], 32, 16, 4, 0.0, n_layers=2, grad_mode='direct',
                  graph=True).to(dev).train()
    red = GradientAllReducer(model.parameters(), model=model)
    for step in range(4):                       # eager + capture on both slots, then replays
        for p in model.parameters():
            p.grad = None
        out, _, _ = model(*dense)
        loss = losses.fused_classification_loss(out, labels, w, dp_global_norm=(mode == 'dist'))
        loss.backward()
        if mode == 'dist':
            flat = model.flat_grad_buffer()
            assert flat is not None, 'graph mode must expose its flat gradient buffer'
            ptr = flat.data_ptr()
            red()
            assert model.flat_grad_buffer().data_ptr() == ptr and model.den1.weight.grad.data_ptr() >= ptr
    torch.cuda.synchronize()
    res[mode] = (float(loss), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
assert abs(res['plain'][0] - res['dist'][0]) <= 1e-6 * max(1.0, abs(res['plain'][0])), (res['plain'][0], res['dist'][0])
for k, g in res['plain'][1].items():
    d = (res['dist'][1][k] - g).abs().max().item()
    assert d <= 1e-6 * max(g.abs().max().item(), 1e-30) + 1e-12, (k, d)
# same shutdown order as tests/dist_multi_check.py: graphs and buffers, then the process group, then out through os._exit
torch.cuda.synchronize()
model.release_graphs()
del model, red
dist.destroy_process_group()
print('DIST_WORLD1_OK', flush=True)
sys.stdout.flush()
sys.stderr.flush()
os._exit(0)
