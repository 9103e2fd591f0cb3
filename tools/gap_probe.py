#!/usr/bin/env python3
"""What do the eager kernels between the two captured graphs of a step cost?  Same training step with (a) the fused
loss (three eager kernels between forward and backward graph), (b) out.backward(g) with g a foreign tensor (one eager
copy), (c) out.backward(g) with g the captured gradient buffer itself (nothing between the graphs)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from eagcn_amd.losses import fused_classification_loss  # noqa: E402
from eagcn_amd.synthetic import bce_weights, make_batch  # noqa: E402

cfg = dict(bench.WORKLOADS['tox21_c2'])
dev = torch.device('cuda', 0)
mb = make_batch(B=cfg['batch'], n_max=cfg['n_max'], n_med=cfg['n_med'], rel_channels=(28, 4, 2, 2, 2), seed=1234, n_tasks=12)
dense = mb.dense(dev)
labels = torch.from_numpy(mb.labels).to(dev)
bw = torch.tensor(bce_weights(12), device=dev)
model = bench.build_model(cfg, 0.3, dev, graph=True).train()
params = list(model.parameters())
g_foreign = torch.randn(cfg['batch'], 12, device=dev) * 1e-3


def step(mode):
    for p in params:
        p.grad = None
    out, _, _ = model(*dense)
    if mode == 'loss':
        fused_classification_loss(out, labels, bw).backward()
    elif mode == 'foreign':
        out.backward(g_foreign)
    else:
        out.backward(out._eagcn_grad_slot)


for mode in ('loss', 'foreign', 'slot', 'loss', 'foreign', 'slot'):
    for _ in range(20):
        step(mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        step(mode)
    torch.cuda.synchronize()
    print('%-8s %.1f us/step' % (mode, (time.perf_counter() - t0) / 300 * 1e6))
