#!/usr/bin/env python3
"""Wall-clock split of one training step on the host (no GPU waits inside the measured spans
except where noted): how long each phase takes to ISSUE."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from eagcn_amd import ops  # noqa: E402
from eagcn_amd.losses import classification_loss  # noqa: E402
from eagcn_amd.synthetic import bce_weights, make_batch  # noqa: E402

cfg = dict(bench.WORKLOADS['tox21_c2'])
dev = torch.device('cuda', 0)
mb = make_batch(B=cfg['batch'], n_max=cfg['n_max'], n_med=cfg['n_med'], rel_channels=(28, 4, 2, 2, 2), seed=1234, n_tasks=12)
dense = mb.dense(dev)
labels = torch.from_numpy(mb.labels).to(dev)
bw = torch.tensor(bce_weights(12), device=dev)
model = bench.build_model(cfg, 0.3, dev).train()
params = list(model.parameters())
acc = {}


def tick(name, t0):
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0)
    return t1


def step(measure):
    t = time.perf_counter()
    for p in params:
        p.grad = None
    if measure: t = tick('zero_grad', t)
    out, _, _ = model(*dense)
    if measure: t = tick('forward (index+sync+engine)', t)
    loss = classification_loss(out, labels, bw)
    if measure: t = tick('loss fwd', t)
    loss.backward()
    if measure: t = tick('backward', t)


for _ in range(20):
    step(False)
torch.cuda.synchronize()
N = 100
t0 = time.perf_counter()
for _ in range(N):
    step(True)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
for k, v in acc.items():
    print('%-32s %8.1f us/step' % (k, v / N * 1e6))
print('%-32s %8.1f us/step' % ('total wall', tot / N * 1e6))
# index alone
t0 = time.perf_counter()
for _ in range(N):
    ops.BatchIndex(dense[0], dense[2:-1])
torch.cuda.synchronize()
print('%-32s %8.1f us' % ('BatchIndex alone (incl. sync)', (time.perf_counter() - t0) / N * 1e6))
