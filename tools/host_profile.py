#!/usr/bin/env python3
"""cProfile of the bench step loop (host side), to see where Python/driver time goes."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from eagcn_amd.losses import fused_classification_loss as classification_loss  # noqa: E402
from eagcn_amd.synthetic import bce_weights, make_batch  # noqa: E402

cfg = dict(bench.WORKLOADS['tox21_c2'])
dev = torch.device('cuda', 0)
import os as _os
mb = make_batch(B=int(_os.environ.get('B', cfg['batch'])), n_max=cfg['n_max'], n_med=cfg['n_med'], rel_channels=(28, 4, 2, 2, 2), seed=1234,
                n_tasks=12)
dense = mb.dense(dev)
labels = torch.from_numpy(mb.labels).to(dev)
bw = torch.tensor(bce_weights(12), device=dev)
model = bench.build_model(cfg, 0.3, dev).train()


params = list(model.parameters())


def step():
    for p in params:
        p.grad = None
    out, _, _ = model(*dense)
    loss = classification_loss(out, labels, bw)
    loss.backward()


for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(24)
