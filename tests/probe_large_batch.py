#!/usr/bin/env python3
"""Is a gradient mismatch at B > 512 (wave-per-tile aggregation path) a defect or fp32 conditioning?  Compares the
HIP gradients and the fp32 oracle's gradients with the fp64 oracle on the same batch.  (Lives in tests/: uses the
oracle as a checker.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from eagcn_amd import EAGCN  # noqa: E402
from eagcn_amd.synthetic import make_batch  # noqa: E402
from oracle.eagcn_ref import RefEAGCN, weights_init_  # noqa: E402

structure, B, n_max = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(5)
w1, w2 = ([80] * 5, [140] * 5) if structure == 'Concate' else ([12] * 5, [20] * 5)
mb = make_batch(B=B, n_max=n_max, n_med=16, rel_channels=(28, 4, 2, 2, 2), seed=11)
ref = RefEAGCN(28, 24, w1, w2, 256, 64, 12, 0.0, structure=structure, n_layers=2)
weights_init_(ref)
ref64 = RefEAGCN(28, 24, w1, w2, 256, 64, 12, 0.0, structure=structure, n_layers=2).double()
ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
hip = EAGCN(28, 24, *w1, *w2, 256, 64, 12, 0.0, structure=structure, n_layers=2).cuda()
hip.load_state_dict(ref.state_dict(), strict=True)
cpu = mb.dense()
gsel = torch.randn(B, 12)
grads = {}
for name, m, d, g in (('ref32', ref, cpu, gsel), ('ref64', ref64, [t.double() if t.is_floating_point() else t for t in cpu], gsel.double()),
                      ('hip', hip, [t.cuda() for t in cpu], gsel.cuda())):
    out, _, _ = m(*d)
    (out * g).sum().backward()
    grads[name] = {k: p.grad.detach().double().cpu() for k, p in m.named_parameters() if p.grad is not None}
rows = []
for k, g64 in grads['ref64'].items():
    den = g64.abs().max().item()
    if den < 1e-9:
        continue
    rows.append(((grads['hip'][k] - g64).abs().max().item() / den, (grads['ref32'][k] - g64).abs().max().item() / den, k))
rows.sort(key=lambda r: -r[0] / max(r[1], 2e-6))           # where HIP is worse than the fp32 oracle
print('%s B=%d N=%d: error vs the fp64 oracle, relative to max|grad| of the parameter' % (structure, B, n_max))
print('%-40s %12s %12s' % ('parameter', 'HIP fp32', 'oracle fp32'))
for eh, er, k in rows[:8]:
    print('%-40s %12.3e %12.3e' % (k, eh, er))
