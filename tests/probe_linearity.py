#!/usr/bin/env python3
"""Noise floor of the backward-linearity property (tests/test_gpu_fullsize.py) for one configuration: how far is
grad(2 g1 - 3 g2) from 2 grad(g1) - 3 grad(g2), per parameter, relative to max|ref|, for the current GEMM path."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))     # this file lives in tests/: it uses the oracle as a checker
import torch  # noqa: E402

import test_gpu_fullsize as T  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'hiv_c3'
c, mb, model = T._setup(name)
dense = [t.cuda() for t in mb.dense()]
B = c['B']
torch.manual_seed(7)
g1, g2 = torch.randn(B, c['nclass'], device='cuda'), torch.randn(B, c['nclass'], device='cuda')


def run(d, cot):
    model.zero_grad(set_to_none=True)
    out, _, gr = model(*d)
    (out * cot).sum().backward()
    return {k: p.grad.detach().double().clone() for k, p in model.named_parameters() if p.grad is not None}


ga, gb, gc = run(dense, g1), run(dense, g2), run(dense, 2.0 * g1 - 3.0 * g2)
ga2 = run(dense, g1)
worst = []
for k in ga:
    want = 2.0 * ga[k] - 3.0 * gb[k]
    e = (gc[k] - want).abs().max().item() / max(want.abs().max().item(), 1e-30)
    rr = (ga2[k] - ga[k]).abs().max().item() / max(ga[k].abs().max().item(), 1e-30)
    worst.append((e, rr, k))
worst.sort(reverse=True)
print('EAGCN_GEMM_X6=%s  worst linearity errors (relative to max|ref|), and run-to-run difference of the same gradient:' % os.environ.get('EAGCN_GEMM_X6', 'default'))
for e, rr, k in worst[:6]:
    print('  %-36s lin %.3e   rerun %.3e' % (k, e, rr))

# ---- against the fp64 oracle on the same batch (small enough: a few seconds on the host)
from helpers import build_oracle_model  # noqa: E402

meta = dict(n_bfeat=c['n_bfeat'], n_afeat=24, widths1=c['w1'], widths2=c['w2'], dens=c['dens'], nclass=c['nclass'],
            structure=c['structure'], molfp='sum')
ref = build_oracle_model(meta, n_layers=c['n_layers']).double().train()
ref.load_state_dict({k: v.double().cpu() for k, v in model.state_dict().items()}, strict=True)
d64 = [t.double() if t.is_floating_point() else t for t in mb.dense()]
out, _, _ = ref(*d64)
(out * g1.double().cpu()).sum().backward()
gr = {k: p.grad.detach().clone() for k, p in ref.named_parameters() if p.grad is not None}
scale = max(v.abs().max().item() for v in gr.values())
rows = []
for k in ga:
    den = max(gr[k].abs().max().item(), 1e-30)
    rows.append(((ga[k].cpu() - gr[k]).abs().max().item() / den, (ga[k].cpu() - gr[k]).abs().max().item() / scale, k))
rows = [r for r in rows if gr[r[2]].abs().max().item() > 1e-6 * scale]          # analytically-zero gradients aside
rows.sort(reverse=True)
print('largest errors of grad(g1) against the fp64 oracle (relative to max|ref| of the parameter / of all parameters):')
for e, es, k in rows[:6] + [r for r in rows if 'ave.weight' in r[2]]:
    print('  %-36s %.3e  %.3e' % (k, e, es))
