"""Graph mode: how long does the host need to ISSUE n steps vs how long until the GPU has finished them?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eagcn_amd.losses import fused_classification_loss
from eagcn_amd.synthetic import bce_weights, make_batch
cfg = dict(bench.WORKLOADS['tox21_c2']); dev = torch.device('cuda', 0)
mb = make_batch(B=256, n_max=132, n_med=16, rel_channels=(28, 4, 2, 2, 2), seed=1234, n_tasks=12)
d = mb.dense(dev); labels = torch.from_numpy(mb.labels).to(dev); bw = torch.tensor(bce_weights(12), device=dev)
for overlap in (False, True):
    model = bench.build_model(cfg, 0.3, dev, graph=True).train(); model.overlap_index = overlap
    params = list(model.parameters())
    def step():
        for p in params: p.grad = None
        out, _, _ = model(*d)
        fused_classification_loss(out, labels, bw).backward()
    for _ in range(30): step()
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('overlap=%s: host issue %.1f us/step, total %.1f us/step' % (overlap, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
