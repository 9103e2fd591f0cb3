"""After tools/flag_stamps_patch.py + rebuild: period of the step graphs and the time their first launch polls for its batch, from device
clock stamps of an UNPROFILED loop.  python tools/flag_stamps.py [batch]"""
import sys, ctypes as C, torch
sys.path.insert(0, '.')
import bench
from eagcn_amd import _lib as L
from eagcn_amd.synthetic import bce_weights, make_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = dict(bench.WORKLOADS['tox21_c2'])
dev = torch.device('cuda', 0)
mb = make_batch(B=B, n_max=cfg['n_max'], n_med=cfg['n_med'], rel_channels=(28, 4, 2, 2, 2), seed=1234, n_tasks=12)
dense = mb.dense(dev)
labels = torch.from_numpy(mb.labels).to(dev)
bw = torch.tensor(bce_weights(12), dtype=torch.float32, device=dev)
model = bench.build_model(cfg, 0.3, dev, graph=True).train()
lib = L.load()
lib.eagcn_debug_flag_stamps.argtypes = [C.c_void_p, C.c_void_p]
for _ in range(200):
    for p in model.parameters():
        p.grad = None
    model.fused_step(dense, labels, 'class', bw, None)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 2048)()
n = C.c_uint()
lib.eagcn_debug_flag_stamps(buf, C.byref(n))
k = n.value
idx = [(k - 60 + i) & 511 for i in range(60)]
st = [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2]) for i in idx]
per = [(st[i + 1][0] - st[i][0]) / 100.0 for i in range(59)]
poll = [(e - s) / 100.0 for s, e, d in st]
body = sorted((d - e) / 100.0 for s, e, d in st)
gap = sorted((st[i + 1][0] - st[i][2]) / 100.0 for i in range(59))
print('graph body (first launch end -> done signal) us: median %.1f min %.1f max %.1f;  gap (done signal -> next first launch) us: median %.1f min %.1f max %.1f' % (body[30], body[0], body[-1], gap[29], gap[0], gap[-1]))
per.sort(); 
print('steps seen %d; period us: median %.1f min %.1f max %.1f; poll us: median %.2f max %.2f' % (k, per[len(per) // 2], per[0], per[-1], sorted(poll)[len(poll) // 2], max(poll)))
print('polls:', ' '.join('%.1f' % p for p in poll[-20:]))
