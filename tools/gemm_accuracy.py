#!/usr/bin/env python3
"""Accuracy of the library GEMM against an fp64 reference, in units of fp32 epsilon relative to sum_k |a||b|
(the natural scale of a dot product's rounding error).  Run with EAGCN_GEMM_X6=0 / 1 to compare the fp32 MFMA tile
with the bf16 x 6 tile; torch.matmul (hipBLASLt fp32) is printed as a yardstick."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from eagcn_amd import ops  # noqa: E402

torch.manual_seed(0)
eps = 2.0 ** -24
for name, ta, tb, sa, sb in (('NN 4809x704x400', False, False, (4809, 400), (400, 704)),
                             ('NT 4809x400x704', False, True, (4809, 704), (400, 704)),
                             ('TN 400x704x4809', True, False, (4809, 400), (4809, 704))):
    for dist in ('normal', 'lognormal-signed'):
        a, b = torch.randn(sa, device='cuda'), torch.randn(sb, device='cuda')
        if dist != 'normal':
            a = a.sign() * torch.exp(3.0 * torch.randn(sa, device='cuda'))
            b = b.sign() * torch.exp(3.0 * torch.randn(sb, device='cuda'))
        A = a.t() if ta else a
        B = b.t() if tb else b
        ref = A.double() @ B.double()
        mag = A.double().abs() @ B.double().abs()
        c = ops.gemm(a, b, ta, tb).double()
        t = (A @ B).double()
        e_lib = ((c - ref).abs() / mag).max().item() / eps
        e_tor = ((t - ref).abs() / mag).max().item() / eps
        print('%-18s %-17s library %.2f eps   torch.matmul %.2f eps   (max |err| / sum|a||b|)' % (name, dist, e_lib, e_tor))
print('EAGCN_GEMM_X6 =', os.environ.get('EAGCN_GEMM_X6', 'default'))
