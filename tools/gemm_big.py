import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eagcn_amd import ops
torch.manual_seed(0)
for (M, N, K) in [(4096, 4096, 4096), (8192, 704, 400), (16384, 704, 400), (4809, 704, 400), (4809, 704, 1600)]:
    a = torch.randn(M, K, device='cuda'); b = torch.randn(K, N, device='cuda')
    for _ in range(3): ops.gemm(a, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n): ops.gemm(a, b)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print('cfg %s  %dx%dx%d: %8.1f us %6.1f TF' % (os.environ.get('EAGCN_GEMM_CFG'), M, N, K, us, 2.0 * M * N * K / us / 1e6))
