"""Phase cycles of one lagg workgroup (after tools/lagg_stamps_patch.py + rebuild): python tools/lagg_stamps.py B n_max n_med"""
import sys, ctypes as C, torch
sys.path.insert(0, '.')
from eagcn_amd import EAGCN, _lib as L
from eagcn_amd.synthetic import make_batch
B, n_max, n_med = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mb = make_batch(B=B, n_max=n_max, n_med=n_med, rel_channels=(28, 4, 2, 2, 2), seed=31, n_tasks=12)
dense = [t.cuda() for t in mb.dense()]
m = EAGCN(28, 24, *[80] * 5, *[140] * 5, 256, 64, 12, 0.3, structure='Concate', n_layers=2, grad_mode='direct').cuda().train()
cot = torch.randn(B, 12, device='cuda')
lib = L.load()
lib.eagcn_debug_lagg_stamps.argtypes = [C.c_void_p]
buf = (C.c_ulonglong * 32)()
for it in range(4):
    for p in m.parameters(): p.grad = None
    out, _, gr = m(*dense)
    (out * cot).sum().backward()
    torch.cuda.synchronize()
    lib.eagcn_debug_lagg_stamps(buf)
    for d in range(2):
        t = [buf[d * 16 + i] for i in range(13)]
        names = ['pro+blk', 'issue', 'B1+stage', 'B2+S+rec', 'B2b+recw', 'B3', 'edge', 'fast', 'slow', 'epi']
        seq = [0, 1, 2, 4, 6, 7, 8, 11, 12, 9, 10] if d else [0, 1, 2, 4, 6, 7, 8, 11, 12, 9, 9]
        print('TRANS' if d else 'FWD  ', ' '.join('%s %5d' % (names[i], t[seq[i + 1]] - t[seq[i]]) for i in range(10)), ' total', t[seq[-1]] - t[0])
