#!/usr/bin/env python3
"""Is the graph-mode training step host- or GPU-bound?  Issue bursts of 3 steps into an EMPTY queue (the host
never blocks on back-pressure) and compare the time the host needs to ISSUE a step with the time the GPU needs
to EXECUTE it (steady-state wall clock per step)."""
import os
import statistics
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
spans = {}


def step(rec=None):
    t = time.perf_counter()
    for p in params:
        p.grad = None
    t1 = time.perf_counter()
    out, _, _ = model(*dense)
    t2 = time.perf_counter()
    loss = fused_classification_loss(out, labels, bw)
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    if rec is not None:
        for k, v in (('zero_grad', t1 - t), ('forward', t2 - t1), ('loss', t3 - t2), ('backward', t4 - t3)):
            rec.setdefault(k, []).append(v)


for _ in range(30):
    step()
torch.cuda.synchronize()
issue = []
for _ in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step(spans)
    issue.append((time.perf_counter() - t0) / 3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 200
print('host issue time per step (empty queue): median %.1f us, min %.1f us' % (statistics.median(issue) * 1e6, min(issue) * 1e6))
for k, v in spans.items():
    print('   %-10s %.1f us' % (k, statistics.median(v) * 1e6))
print('steady-state wall per step: %.1f us' % (wall * 1e6))
