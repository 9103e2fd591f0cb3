#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM on the three product shapes of a layer, per tile config.
Usage: EAGCN_GEMM_CFG=<id> python tools/gemm_bench.py   (one process per config: the choice is read once)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from eagcn_amd import ops  # noqa: E402

T = int(os.environ.get('T', 4809))
shapes = [('fwd  NN', False, False, (T, 400), (400, 704)),
          ('dX   NT', False, True, (T, 704), (400, 704)),
          ('dW   TN', True, False, (T, 400), (T, 704)),
          ('l1fwd NN', False, False, (T, 24), (24, 400))]
torch.manual_seed(0)
res = []
for name, ta, tb, sa, sb in shapes:
    a = torch.randn(sa, device='cuda')
    b = torch.randn(sb, device='cuda')
    for _ in range(5):
        ops.gemm(a, b, ta, tb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        ops.gemm(a, b, ta, tb)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    M = sa[1] if ta else sa[0]
    K = sa[0] if ta else sa[1]
    N = sb[0] if tb else sb[1]
    res.append('%s %7.1f us %6.1f TF' % (name, us, 2.0 * M * N * K / us / 1e6))
print('cfg %s: ' % os.environ.get('EAGCN_GEMM_CFG', 'auto') + ' | '.join(res))
