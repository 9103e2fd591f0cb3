"""Engine vs layer-wise composition vs a second engine run at a full-size config: largest gradient differences per tensor
(diagnostic for tests/test_gpu_fullsize.py).   python tools/diff_paths.py hiv_c3"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from test_gpu_fullsize import _grads, _setup

name = sys.argv[1] if len(sys.argv) > 1 else 'hiv_c3'
c, mb, a = _setup(name, grad_mode='direct')
dense = list(mb.dense('cuda'))
cot = torch.randn(c['B'], c['nclass'], device='cuda')
res = []
for fn in (a.forward, a.forward, a.forward_composed, a.forward_composed):
    for p in a.parameters():
        p.grad = None
    out, _, gr = fn(*dense)
    (out * cot).sum().backward()
    res.append((out.detach().clone(), _grads(a), gr.detach().clone()))
scale = max(v.abs().max().item() for v in res[0][1].values())
print('graph_rep: composition vs engine %.2e, composition run 2 vs run 1 %.2e' % ((res[2][2] - res[0][2]).abs().max().item() / res[0][2].abs().max().item(), (res[3][2] - res[2][2]).abs().max().item() / res[0][2].abs().max().item()))
for tag, (o, g, _) in (('engine run 2', res[1]), ('composition', res[2]), ('composition run 2', res[3])):
    rows = []
    base = res[2] if tag == 'composition run 2' else res[0]
    for k, v in base[1].items():
        d = (g[k] - v).abs().max().item()
        rows.append((d / scale, d / max(v.abs().max().item(), 1e-30), k))
    rows.sort(reverse=True)
    print('%s vs %s: out diff %.2e; worst tensors (err / case scale, err / own max):' % (tag, 'composition run 1' if tag == 'composition run 2' else 'engine run 1', (o - base[0]).abs().max().item() / base[0].abs().max().item()))
    for r in rows[:5]:
        print('   %.2e  %.2e  %s' % r)
