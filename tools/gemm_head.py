import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eagcn_amd import ops
torch.manual_seed(0)
res = []
for name, ta, tb, sa, sb in [('den1 fwd', 0, 0, (256, 700), (700, 256)), ('den1 dX', 0, 1, (256, 256), (700, 256)),
                             ('den1 dW', 1, 0, (256, 700), (256, 256)), ('den2 fwd', 0, 0, (256, 256), (256, 64)),
                             ('den3 fwd', 0, 0, (256, 64), (64, 12)), ('l1 fwd', 0, 0, (4809, 24), (24, 400)),
                             ('l1 dW', 1, 0, (4809, 24), (4809, 400))]:
    a = torch.randn(sa, device='cuda'); b = torch.randn(sb, device='cuda')
    for _ in range(5): ops.gemm(a, b, bool(ta), bool(tb))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): ops.gemm(a, b, bool(ta), bool(tb))
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    res.append('%s %.1f' % (name, e0.elapsed_time(e1) / 20 * 1e3))
print('cfg', os.environ.get('EAGCN_GEMM_CFG', 'auto'), ' | '.join(res))
