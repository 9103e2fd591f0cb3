#!/usr/bin/env python3
"""Single layer, forward + backward, HIP vs the fp64 oracle for batch sizes around 1024 (lives in tests/: uses the
oracle as a checker)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402

from eagcn_amd import GraphConv_Layer  # noqa: E402
from eagcn_amd.synthetic import make_batch  # noqa: E402
from helpers import rel_err  # noqa: E402
from oracle.eagcn_ref import RefGraphConvLayer  # noqa: E402

widths = [int(x) for x in os.environ.get('W', '80,80,80,80,80').split(',')]
for B in [int(x) for x in sys.argv[1:]]:
    torch.manual_seed(1)
    mb = make_batch(B=B, n_max=12, n_med=16, rel_channels=(28, 4, 2, 2, 2), seed=11)
    ref = RefGraphConvLayer(24, (28, 4, 2, 2, 2), widths, 0.0, 'Concate').double()
    for p in ref.parameters():
        if p.dim() > 1:
            torch.nn.init.normal_(p, 0.0, 0.3)
    hip = GraphConv_Layer(24, 28, *widths, 0.0, 'Concate')
    hip.load_state_dict({k: v.float() for k, v in ref.state_dict().items()}, strict=True)
    hip.cuda().train()
    cpu = mb.dense()
    adj, afm, rels = cpu[0], cpu[1], cpu[2:-1]
    x64 = afm.double().requires_grad_(True)
    y_r, _ = ref(adj.double(), x64, *[r.double() for r in rels])
    xh = afm.cuda().requires_grad_(True)
    y_h, _ = hip(adj.cuda(), xh, *[r.cuda() for r in rels])
    g = torch.randn(y_r.shape, dtype=torch.float64)
    (y_r * g).sum().backward()
    (y_h * g.float().cuda()).sum().backward()
    print('B=%d  y %.2e  dx %.2e' % (B, rel_err(y_h.detach().cpu(), y_r.detach()), rel_err(xh.grad.cpu(), x64.grad)))
    pr, ph = dict(ref.named_parameters()), dict(hip.named_parameters())
    rows = sorted(((rel_err(ph[k].grad.cpu(), v.grad), k) for k, v in pr.items() if v.grad is not None and v.grad.abs().max() > 1e-9), reverse=True)
    print('   ', ['%s %.1e' % (k, e) for e, k in rows[:5]])
