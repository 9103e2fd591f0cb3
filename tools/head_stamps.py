"""Phase cycles of one head_mid workgroup (after tools/head_stamps_patch.py + rebuild): python tools/head_stamps.py"""
import sys, ctypes as C, torch
sys.path.insert(0, '.')
import bench
from eagcn_amd import _lib as L
from eagcn_amd.synthetic import bce_weights, make_batch
cfg = dict(bench.WORKLOADS['tox21_c2'])
dev = torch.device('cuda', 0)
mb = make_batch(B=256, n_max=cfg['n_max'], n_med=cfg['n_med'], rel_channels=(28, 4, 2, 2, 2), seed=1234, n_tasks=12)
dense = mb.dense(dev)
labels = torch.from_numpy(mb.labels).to(dev)
bw = torch.tensor(bce_weights(12), dtype=torch.float32, device=dev)
model = bench.build_model(cfg, 0.3, dev, graph=True).train()
lib = L.load()
lib.eagcn_debug_head_stamps.argtypes = [C.c_void_p]
buf = (C.c_ulonglong * 16)()
names = ['stage loads', 'loss loads', 'tables', 'sync', 'F3 tile', 'store+sync', 'loss+sync', 'B3a tiles']
for it in range(6):
    for p in model.parameters():
        p.grad = None
    model.fused_step(dense, labels, 'class', bw, None)
    torch.cuda.synchronize()
    lib.eagcn_debug_head_stamps(buf)
    t = [buf[i] for i in range(9)]
    print(' '.join('%s %5d' % (names[i], t[i + 1] - t[i]) for i in range(8)), ' total', t[8] - t[0])
