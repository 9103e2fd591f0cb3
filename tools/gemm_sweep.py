#!/usr/bin/env python3
"""Where does the fp32 MFMA GEMM lose time at the layer shapes?  K sweep (slope = cost per k-tile, intercept =
launch + prologue + epilogue) and M sweep (workgroup-count quantisation over the 256 CUs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from eagcn_amd import ops  # noqa: E402


def t_us(M, N, K, ta=False, tb=False, n=40):
    a = torch.randn((K, M) if ta else (M, K), device='cuda')
    b = torch.randn((N, K) if tb else (K, N), device='cuda')
    for _ in range(5):
        ops.gemm(a, b, ta, tb)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gemm(a, b, ta, tb)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print('K sweep, M=4809 N=704 (836 tiles), NN:')
for K in (16, 64, 128, 256, 400, 800, 1600, 3200):
    us = t_us(4809, 704, K)
    print('  K=%5d  %7.1f us  %6.1f TF' % (K, us, 2.0 * 4809 * 704 * K / us / 1e6))
print('K sweep, M=4809 N=400 (532 tiles), NT:')
for K in (64, 256, 704, 1408, 2816):
    us = t_us(4809, 400, K, False, True)
    print('  K=%5d  %7.1f us  %6.1f TF' % (K, us, 2.0 * 4809 * 400 * K / us / 1e6))
print('M sweep, N=704 K=400, NN (tiles = 11 * ceil(M/64)):')
for mt in (23, 24, 46, 47, 69, 70, 76, 93, 94, 116, 117, 186):
    M = 64 * mt
    us = t_us(M, 704, 400)
    print('  M=%5d tiles=%5d (%.2f per CU)  %7.1f us  %6.1f TF' % (M, 11 * mt, 11 * mt / 256.0, us, 2.0 * M * 704 * 400 / us / 1e6))
