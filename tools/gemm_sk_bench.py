#!/usr/bin/env python3
"""Correctness + timing of the wave-autonomous balanced GEMM (csrc/gemm3.hip, forms NT / TN) on the product shapes of a
layer, next to the one-workgroup-per-tile kernel of csrc/gemm.hip.  One process per configuration (EAGCN_GEMM3_WGS is
read once):    T=4809 FIN=400 FP=704 python tools/gemm_sk_bench.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from eagcn_amd import _lib, ops  # noqa: E402

lib = _lib.load()
F_WS, F_TO, F_GEMM, F_PAIR = lib.eagcn_gemm_sk_workspace_bytes, lib.eagcn_gemm_sk_timeouts, lib.eagcn_gemm_f32_sk, lib.eagcn_gemm_pair_sk
T = int(os.environ.get('T', 4809))
FIN, FP = int(os.environ.get('FIN', 400)), int(os.environ.get('FP', 704))
CHECK = os.environ.get('CHECK', '1') == '1'
ws = torch.empty(F_WS(), dtype=torch.uint8, device='cuda')
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.manual_seed(0)


def sk(ta, tb, a, b):
    M = a.shape[1] if ta else a.shape[0]
    K = a.shape[0] if ta else a.shape[1]
    N = b.shape[0] if tb else b.shape[1]
    c = torch.empty((M, N), device='cuda')
    _lib.check(F_GEMM(int(ta), int(tb), M, N, K, a.data_ptr(), a.shape[1], b.data_ptr(), b.shape[1],
                                     c.data_ptr(), N, ws.data_ptr(), ws.numel(), s), 'sk')
    return c


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def err(c, ref):
    return ((c.double() - ref).abs().max() / ref.abs().max()).item()


x = torch.randn(T, FIN, device='cuda')
w = torch.randn(FIN, FP, device='cuda')
dp = torch.randn(T, FP, device='cuda')
res = []
wt = w.t().contiguous()
# (the wave-autonomous kernel runs the forward transform in the NT form on the pre-transposed weight)
shapes = [('fwd NT', False, True, x, wt),
          ('dX NT', False, True, dp, w), ('dW TN', True, False, x, dp)]
for name, ta, tb, a, b in shapes:
    c = sk(ta, tb, a, b)
    e = -1.0
    if CHECK:
        ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
        e = err(c, ref)
    us = timeit(lambda: sk(ta, tb, a, b))
    us_old = timeit(lambda: ops.gemm(a, b, ta, tb)) if name != 'dW TN' else float('nan')     # (no split-K through this entry)
    fl = 2.0 * T * FIN * FP
    res.append('%s err %.1e %7.1f us %6.1f TF (old %7.1f us)' % (name, e, us, fl / us / 1e6, us_old))
# the pair
dx = torch.empty(T, FIN, device='cuda')
dw = torch.empty(FIN, FP, device='cuda')


def pair():
    _lib.check(F_PAIR(T, FIN, FP, dp.data_ptr(), FP, w.data_ptr(), FP, dx.data_ptr(), FIN,
                                      FIN, FP, T, x.data_ptr(), FIN, dp.data_ptr(), FP, dw.data_ptr(), FP,
                                      ws.data_ptr(), ws.numel(), s), 'pair')


pair()
e1 = e2 = -1.0
if CHECK:
    e1 = err(dx, dp.double() @ w.double().t())
    e2 = err(dw, x.double().t() @ dp.double())
us = timeit(pair)
res.append('pair err %.1e/%.1e %7.1f us %6.1f TF' % (e1, e2, us, 4.0 * T * FIN * FP / us / 1e6))
# the pair on the XCD-local schedule (8 partial dW slabs + their sum, which the step's unpack_grads launch performs)
slabs = torch.empty(8, FIN, FP, device='cuda')


def pair_xk():
    _lib.check(lib.eagcn_gemm_pair_sk_slabs(T, FIN, FP, dp.data_ptr(), FP, w.data_ptr(), FP, dx.data_ptr(), FIN,
                                            FIN, FP, T, x.data_ptr(), FIN, dp.data_ptr(), FP, slabs.data_ptr(), FP,
                                            FIN * FP, ws.data_ptr(), ws.numel(), s), 'pair_xk')


dx.zero_()
pair_xk()
if CHECK:
    e1 = err(dx, dp.double() @ w.double().t())
    e2 = err(slabs.sum(0), x.double().t() @ dp.double())
us = timeit(pair_xk)
us_sum = timeit(lambda: slabs.sum(0))
res.append('pair XCD-local err %.1e/%.1e %7.1f us incl. slab memset (%6.1f TF), slab sum %.1f us' % (e1, e2, us, 4.0 * T * FIN * FP / us / 1e6, us_sum))
# the same launch with COLD caches (a 768 MB read-modify-write between the calls evicts L2 and the 256 MB MALL), and with
# a small kernel stream in front of it as in a training step (launch right behind other work instead of back to back)
big = torch.zeros(192 << 20, device='cuda')
small = torch.zeros(1 << 16, device='cuda')
for label, pre in (('cold', lambda: big.add_(1.0)), ('behind small kernels', lambda: [small.add_(1.0) for _ in range(4)])):
    ts = []
    for it in range(12):
        pre()
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea.record()
        pair()
        eb.record()
        torch.cuda.synchronize()
        if it >= 2:
            ts.append(ea.elapsed_time(eb) * 1e3)
    ts.sort()
    res.append('pair %s: median %.1f us (min %.1f)' % (label, ts[len(ts) // 2], ts[0]))
# ragged / tiny shapes (correctness only): K tail of 8, M not a multiple of 64, N = 400, fewer iterations than workgroups
if CHECK:
    bad = []
    for (M, N, K) in [(100, 64, 24), (1000, 400, 24), (37, 16, 8), (4809, 400, 24), (133, 72, 100), (5000, 704, 400)]:
        for ta, tb in [(0, 1), (1, 0)]:
            Mq = (M + 3) // 4 * 4 if ta else M
            a = torch.randn((K, Mq) if ta else (Mq, K), device='cuda')
            b = torch.randn((N, K) if tb else (K, N), device='cuda')
            if ta:
                K2 = K
            c = sk(ta, tb, a, b)
            ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
            e = err(c, ref)
            if not e < 2e-6:
                bad.append((M, N, K, ta, tb, e))
    res.append('ragged: %s' % ('ok' if not bad else bad))
torch.cuda.synchronize()
print('T=%d FIN=%d FP=%d wgs=%s timeouts=%d | ' % (T, FIN, FP, os.environ.get('EAGCN_GEMM3_WGS', '256'), F_TO()) + ' | '.join(res))
