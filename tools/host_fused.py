#!/usr/bin/env python3
"""Is the fused training step of bench.py limited by the host or by the GPU?  Host ISSUE time per step (the Python call
returns as soon as everything is enqueued) against the wall-clock per step, plus a cProfile of the issue path."""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from eagcn_amd.synthetic import bce_weights, make_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--steps', type=int, default=300)
ap.add_argument('--profile', action='store_true')
a = ap.parse_args()
cfg = dict(bench.WORKLOADS['tox21_c2'])
dev = torch.device('cuda', 0)
mb = make_batch(B=a.batch, n_max=cfg['n_max'], n_med=cfg['n_med'], rel_channels=(28, 4, 2, 2, 2), seed=1234, n_tasks=12)
dense = mb.dense(dev)
labels = torch.from_numpy(mb.labels).to(dev)
bw = torch.tensor(bce_weights(12), dtype=torch.float32, device=dev)
model = bench.build_model(cfg, 0.3, dev, graph=True).train()
params = list(model.parameters())


def step():
    for p in params:
        p.grad = None
    return model.fused_step(dense, labels, 'class', bw, None)[0]


for _ in range(20):
    step()
torch.cuda.synchronize()
issue = 0.0
t_all = time.perf_counter()
for _ in range(a.steps):
    t0 = time.perf_counter()
    step()
    issue += time.perf_counter() - t0
t_loop = time.perf_counter() - t_all
torch.cuda.synchronize()
t_wall = time.perf_counter() - t_all
print('issue %.1f us/step   loop %.1f us/step   wall (with final sync) %.1f us/step   drain after the loop %.1f us'
      % (issue / a.steps * 1e6, t_loop / a.steps * 1e6, t_wall / a.steps * 1e6, (t_wall - t_loop) * 1e6))
if a.profile:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('cumulative').print_stats(28)
