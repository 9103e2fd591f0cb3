"""Probe: per-kernel-class time of the eager engine as a function of the row capacity."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from eagcn_amd import _lib, ops
from eagcn_amd.losses import fused_classification_loss
from eagcn_amd.synthetic import bce_weights, make_batch
lib = _lib.load()
cfg = dict(bench.WORKLOADS['tox21_c2'])
dev = torch.device('cuda', 0)
mb = make_batch(B=256, n_max=132, n_med=16, rel_channels=(28, 4, 2, 2, 2), seed=1234, n_tasks=12)
d = mb.dense(dev)
labels = torch.from_numpy(mb.labels).to(dev)
bw = torch.tensor(bce_weights(12), device=dev)
model = bench.build_model(cfg, 0.3, dev, graph=False).train()
params = list(model.parameters())
T = int(mb.sizes.sum())
for cap in (None, T + 16, 2 * T, 4 * T, 256 * 132):
    orig = ops.BatchIndex.__init__
    def patched(self, adj, rels, overlap=False, row_cap=None, _cap=cap):
        orig(self, adj, rels, overlap=overlap, row_cap=_cap)
    ops.BatchIndex.__init__ = patched
    def step():
        for p in params: p.grad = None
        out, _, _ = model(*d)
        fused_classification_loss(out, labels, bw).backward()
    for _ in range(5): step()
    torch.cuda.synchronize()
    lib.eagcn_prof_reset(); lib.eagcn_prof_enable(1)
    n = 20
    for _ in range(n): step()
    torch.cuda.synchronize(); lib.eagcn_prof_enable(0)
    res = {}
    for tag in range(lib.eagcn_prof_ntags()):
        ms, w, k = C.c_double(), C.c_double(), C.c_int64()
        lib.eagcn_prof_read(tag, C.byref(ms), C.byref(w), C.byref(k))
        res[lib.eagcn_prof_tag_name(tag).decode()] = round(ms.value / n * 1e3, 1)
    print('cap', cap, res, 'sum', round(sum(res.values()), 1))
    ops.BatchIndex.__init__ = orig
