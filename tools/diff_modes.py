"""Gradients of a full-size config under gemm mode 0 and mode 3, engine and layer-wise composition, all against the mode-0 engine.
   python tools/diff_modes.py hiv_c3"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from eagcn_amd import _lib
from test_gpu_fullsize import _grads, _setup

name = sys.argv[1] if len(sys.argv) > 1 else 'hiv_c3'
lib = _lib.load()
res = {}
for mode in (0, 3):
    lib.eagcn_set_gemm_mode(mode)
    c, mb, a = _setup(name, grad_mode='direct')
    dense = list(mb.dense('cuda'))
    torch.manual_seed(123)
    cot = torch.randn(c['B'], c['nclass'], device='cuda')
    for tag, fn in (('engine', a.forward), ('composed', a.forward_composed)):
        for p in a.parameters():
            p.grad = None
        out, _, gr = fn(*dense)
        (out * cot).sum().backward()
        res[(mode, tag)] = (out.detach().clone(), _grads(a), gr.detach().clone())
    del a
for key in ((0, 'composed'), (3, 'composed')):
    base = res[(key[0], 'engine')]
    scale = max(v.abs().max().item() for v in base[1].values())
    o, g, gr = res[key]
    rows = sorted(((g[k] - v).abs().max().item() / scale, (g[k] - v).abs().max().item() / max(v.abs().max().item(), 1e-30), k) for k, v in base[1].items())[::-1]
    print('mode %d %s vs the same mode\'s engine: out %.2e graph_rep %.2e; worst (err / case scale, err / own max):' % (
        key[0], key[1], (o - base[0]).abs().max().item() / base[0].abs().max().item(), (gr - base[2]).abs().max().item() / base[2].abs().max().item()))
    for r in rows[:4]:
        print('   %.2e  %.2e  %s' % r)
    print('   head: ' + ', '.join('%s %.1e' % (r[2], r[1]) for r in rows if not r[2].startswith('layer')))
