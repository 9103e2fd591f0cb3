"""The plane GEMM of gemm mode 3 (csrc/gemm_bx3.hip; reference layers.py:40 `torch.mm` and its two autograd products) through its
C-ABI entry points: exactness of the three-way bf16 split, both operand forms against float64 (edges: extents that are not tile
multiples, k tails, empty / single k-chunks, one launch for the dX / dW pair), the drift of long same-sign reductions, and the
whole model in mode 3 against mode 0 (fp32 MFMA) at widths where every hidden layer runs on it."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from eagcn_amd import _lib as L
    return L.load(), L


@pytest.fixture(params=[0, 1], ids=['tile128x128', 'tile256x128'])
def wide(request):
    """Both kernels behind the entry points (csrc/bx3.h): gemm_bx3.hip (4 compute + 2 loader waves) and gemm_bx3w.hip (8 compute waves,
    two per SIMD, staggered); the library picks by launch size, the tests force each."""
    lib, _ = _lib()
    old = lib.eagcn_set_bx3_wide(request.param)
    yield request.param
    lib.eagcn_set_bx3_wide(old)


def _planes(x, np_=3, spare_rows=5):
    """Plane images (panel-major, include/eagcn_hip.h) of a contiguous fp32 CUDA matrix [R, ld] -> (int16 tensor, plane stride, row
    capacity).  The images are pre-filled with bf16 NaNs and given a few spare rows: whatever the split does not write (spare rows,
    the columns between ld and the end of the last 32-column panel) must never reach a stored result."""
    lib, L = _lib()
    R, ld = x.shape
    cap = R + spare_rows
    stride = (int(lib.eagcn_bx3_plane_elems(cap, ld)) + 63) // 64 * 64
    pl = torch.full((np_ * stride,), 0x7FC0, dtype=torch.int16, device='cuda')
    L.check(lib.eagcn_bx3_split(C.c_void_p(x.data_ptr()), R, ld, C.c_void_p(pl.data_ptr()), stride, cap, np_, None), 'eagcn_bx3_split')
    return pl, stride, cap


def _image_index(R, ld, cap):
    """element index of (r, c) inside a plane image (the formula of include/eagcn_hip.h), as an [R, ld] LongTensor"""
    r = torch.arange(R, device='cuda').view(-1, 1)
    c = torch.arange(ld, device='cuda').view(1, -1)
    return ((c >> 5) * cap + r) * 32 + ((((c >> 3) & 3) ^ ((r >> 2) & 3)) << 3) + (c & 7)


def _bf16_to_f32(p):
    return (p.to(torch.int32) << 16).view(torch.float32)


def test_split_is_exact():
    torch.manual_seed(0)
    x = torch.randn(37, 64, device='cuda') * torch.logspace(-20, 20, 64, device='cuda')
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.17549435e-38, 2.0 ** -100], device='cuda')
    pl, stride, cap = _planes(x)
    idx = _image_index(x.shape[0], x.shape[1], cap)
    parts = [_bf16_to_f32(pl[q * stride:(q + 1) * stride][idx]).double() for q in range(3)]
    assert torch.equal((parts[0] + parts[1] + parts[2]).float(), x), 'x0 + x1 + x2 must reproduce x bit for bit'
    # every piece is the bf16 nearest to what is left: |x1| <= 2^-8 |x0|, |x2| <= 2^-8 |x1|
    assert (parts[1].abs() <= parts[0].abs() * 2.0 ** -8 + 1e-300).all() and (parts[2].abs() <= parts[1].abs() * 2.0 ** -8 + 1e-300).all()
    one, _, cap1 = _planes(x, 1)
    assert torch.equal(_bf16_to_f32(one[_image_index(x.shape[0], x.shape[1], cap1)]), x.to(torch.bfloat16).float())      # round to nearest even
    written = torch.zeros(one.numel(), dtype=torch.bool, device='cuda')
    written[_image_index(x.shape[0], x.shape[1], cap1).reshape(-1)] = True
    assert (one[~written] == 0x7FC0).all(), 'the split writes the elements of the matrix and nothing else'


def _gemm(tn, A, B, M, N, K, splits=1):
    lib, L = _lib()
    pa, sa, ra = _planes(A)
    pb, sb, rb = _planes(B)
    ldc = (N + 3) // 4 * 4
    slab = M * ldc
    Cm = torch.full((max(splits, 1) * slab,), float('nan'), device='cuda')
    L.check(lib.eagcn_gemm_bx3(tn, M, N, K, C.c_void_p(pa.data_ptr()), sa, A.shape[1], ra, C.c_void_p(pb.data_ptr()), sb, B.shape[1], rb,
                               C.c_void_p(Cm.data_ptr()), ldc, splits, slab, 3, None), 'eagcn_gemm_bx3')
    used = lib.eagcn_bx3_used_splits(splits, M, N, K) if tn else 1
    return Cm.view(max(splits, 1), M, ldc)[:used, :, :N].double().sum(0), Cm.view(max(splits, 1), M, ldc)


def _err(got, A64, B64, tn):
    ref = A64.t() @ B64 if tn else A64 @ B64.t()
    den = A64.abs().t() @ B64.abs() if tn else A64.abs() @ B64.abs().t()
    return ((got - ref).abs() / den.clamp_min(1e-300)).max().item()


@pytest.mark.parametrize('M,N,K,pad', [(128, 128, 32, 0), (100, 90, 48, 8), (4809, 720, 400, 16), (37, 16, 128, 0), (300, 200, 24, 8),
                                       (2500, 1264, 512, 0), (1, 1, 8, 0), (77000, 130, 72, 0), (257, 129, 40, 0)])
def test_nt_product_vs_float64(M, N, K, pad, wide):
    """C = A.B^T (forward P = X.Wcat^T and dX = dP.Wcat: both operands K-contiguous); K a multiple of 8, leading dimensions larger than
    K (whatever follows the k range in a row must not reach a result)."""
    torch.manual_seed(M + N + K)
    A = torch.full((M, K + pad), 7.0e30, device='cuda')
    B = torch.full((N, K + pad), -3.0e30, device='cuda')
    A[:, :K] = torch.randn(M, K, device='cuda')
    B[:, :K] = torch.randn(N, K, device='cuda') * 0.05
    got, raw = _gemm(0, A, B, M, N, K)
    e = _err(got, A[:, :K].double(), B[:, :K].double(), False)
    assert e <= 8 * 2.0 ** -24, e
    assert torch.isnan(raw[0, :, N:]).all(), 'columns beyond N were written'


@pytest.mark.parametrize('M,N,K,splits', [(128, 128, 32, 1), (100, 90, 75, 1), (400, 720, 4809, 6), (400, 720, 4809, 1), (512, 1024, 3000, 3),
                                          (128, 16, 37, 2), (64, 64, 700, 64), (8, 8, 1, 1), (300, 260, 33000, 40), (257, 129, 33, 1)])
def test_tn_product_vs_float64(M, N, K, splits, wide):
    """C = A^T.B (dW = X^T.dP: the reduction runs over the packed rows) as k-chunk slabs; rows beyond K read as zero; slabs beyond
    eagcn_bx3_used_splits are not written."""
    lib, _ = _lib()
    torch.manual_seed(M * 3 + N + K)
    A = torch.randn(K, (M + 7) // 8 * 8, device='cuda')
    B = torch.randn(K, (N + 7) // 8 * 8 + 8, device='cuda')
    got, raw = _gemm(1, A, B, M, N, K, splits)
    e = _err(got, A[:, :M].double(), B[:, :N].double(), True)
    assert e <= 8 * 2.0 ** -24, e
    used = lib.eagcn_bx3_used_splits(splits, M, N, K)
    assert 1 <= used <= splits and torch.isnan(raw[used:]).all() and not torch.isnan(raw[:used, :, :N]).any()


def test_pair_launch_equals_the_two_products(wide):
    """dX + dW in one persistent launch: dX bit for bit what the separate launch computes; the weight gradient is cut into k-chunks
    sized against the dX units of the same launch (eagcn_bx3_pair_used_splits), so its bits depend on the cut -- held to the float64
    product like every other TN case, and the slabs beyond the used count stay untouched."""
    lib, L = _lib()
    torch.manual_seed(5)
    T, FIN, FP, splits = 3000, 256, 400, 12
    X = torch.randn(T, FIN, device='cuda').relu()
    dP = torch.randn(T, FP, device='cuda')
    W = torch.randn(FIN, FP, device='cuda') * 0.05
    dx_ref, _ = _gemm(0, dP, W, T, FIN, FP)
    px, sx, rx = _planes(X)
    pp, sp, rp = _planes(dP)
    pw, sw, rw = _planes(W)
    dX = torch.zeros(T, FIN, device='cuda')
    dW = torch.full((splits, FIN, FP), float('nan'), device='cuda')
    L.check(lib.eagcn_gemm_bx3_pair(T, FIN, FP, C.c_void_p(pp.data_ptr()), sp, FP, rp, C.c_void_p(pw.data_ptr()), sw, FP, rw, C.c_void_p(dX.data_ptr()), FIN,
                                    FIN, FP, T, C.c_void_p(px.data_ptr()), sx, FIN, rx, C.c_void_p(pp.data_ptr()), sp, FP, rp, C.c_void_p(dW.data_ptr()), FP,
                                    splits, FIN * FP, 3, None), 'eagcn_gemm_bx3_pair')
    used = lib.eagcn_bx3_pair_used_splits(splits, FIN, FP, T, T, FIN, FP)
    assert torch.equal(dX.double(), dx_ref), 'a dX unit computes the same bits in either launch'
    assert 1 <= used <= splits and torch.isnan(dW[used:]).all() and not torch.isnan(dW[:used]).any()
    e = _err(dW[:used].double().sum(0), X.double(), dP.double(), 1)
    print('pair launch: %d of %d k-chunk slabs used; dW error %.2e of sum |a||b|' % (used, splits, e))
    assert e <= 8 * 2.0 ** -24, e


def test_long_same_sign_reduction_does_not_drift(wide):
    """The bf16 MFMA accumulate is not round-to-nearest; the kernel keeps the exact leading products apart from the five small ones
    (gemm_bx3.hip): a 4096-row chunk -- the longest chain the layer path cuts -- stays within 2e-8 of the float64 sum on average."""
    torch.manual_seed(9)
    K = 4096
    A = torch.rand(K, 128, device='cuda')
    B = torch.rand(K, 128, device='cuda')
    got, _ = _gemm(1, A, B, 128, 128, K, 1)
    ref = A.double().t() @ B.double()
    rel = (got - ref) / ref
    assert abs(rel.mean().item()) <= 2e-8 and rel.abs().max().item() <= 3e-6, (rel.mean().item(), rel.abs().max().item())


@pytest.mark.parametrize('structure', ['Concate', 'Weighted_sum'])
def test_model_in_plane_mode_against_fp32_mfma_mode(structure, wide):
    """Same model, same batch: gemm mode 3 (the default) against mode 0 (fp32 MFMA): outputs to 1e-5, every gradient to 1e-5 of its
    own largest entry + 2e-6 of the case's (the two differ in the rounding of the layer products only)."""
    from eagcn_amd import EAGCN
    from eagcn_amd.synthetic import make_batch
    lib, _ = _lib()
    mb = make_batch(B=48, n_max=60, n_med=16, rel_channels=(28, 4, 2, 2, 2), seed=21)
    dense = [t.cuda() for t in mb.dense()]
    res = {}
    for mode in (0, 3):
        old = lib.eagcn_set_gemm_mode(mode)
        try:
            torch.manual_seed(3)
            m = EAGCN(28, 24, *[48] * 5, *[64] * 5, 64, 32, 3, 0.0, structure=structure, n_layers=3, grad_mode='direct').cuda().train()
            # the head's relus never gate (BatchNorm biases at +6, as tests/test_gpu_fullsize.py): with 48 molecules ONE head unit whose
            # pre-activation sits within rounding of zero flips its gate between two arithmetically equivalent paths and moves the
            # gradients of everything below it by 1/48 (seen when the aggregation kernel changed, round 5: den1 / Graph_BN / every
            # ave.weight off by 1-5 % with outputs equal to 1e-6 and every saved activation equal to rounding)
            with torch.no_grad():
                m.bn_den1.bias.fill_(6.0)
                m.bn_den2.bias.fill_(6.0)
            torch.manual_seed(4)
            cot = torch.randn(48, 3, device='cuda')
            out, _, gr = m(*dense)
            ((out * cot).sum() + 0.1 * gr.sum()).backward()
            res[mode] = (out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        finally:
            lib.eagcn_set_gemm_mode(old)
    o0, g0 = res[0]
    o3, g3 = res[3]
    assert ((o3 - o0).abs().max() / o0.abs().max()).item() < 1e-5
    scale = max(v.abs().max().item() for v in g0.values())
    for k, v in g0.items():
        d = (g3[k] - v).abs().max().item()
        assert d <= 1e-5 * v.abs().max().item() + 2e-6 * scale, (k, d, v.abs().max().item(), scale)


def test_bond_list_aggregation_matches_the_dense_kernels():
    """csrc/lagg.hip forced everywhere (EAGCN_AGG=lds) against the matrix-core kernels forced everywhere (EAGCN_AGG=dense): the
    aggregation as a gather over the bond lists plus ONE rank-one term per molecule is the same operator as the dense block of
    agg.hip (reference layers.py:82-92), exact including the 1e-9 filler -- for both layer structures, whatever the default policy
    picks per direction.  Run in subprocesses (the policy is read once per process): outputs and every gradient."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import sys, torch
sys.path.insert(0, %r)
from eagcn_amd import EAGCN
from eagcn_amd.synthetic import make_batch
mb = make_batch(B=24, n_max=70, n_med=25, rel_channels=(28, 4, 2, 2, 2), seed=31, isolated_frac=0.05)
dense = [t.cuda() for t in mb.dense()]
torch.manual_seed(3)
m = EAGCN(28, 24, *[48] * 5, *[64] * 5, 64, 32, 3, 0.0, structure=sys.argv[1], n_layers=2, grad_mode='direct').cuda().train()
torch.manual_seed(4)
cot = torch.randn(24, 3, device='cuda')
out, _, gr = m(*dense)
((out * cot).sum() + 0.1 * gr.sum()).backward()
torch.save({'out': out.detach().cpu(), 'g': {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None}}, sys.argv[2])
''' % root
    import tempfile
    res = {}
    with tempfile.TemporaryDirectory() as d:
        for structure in ('Concate', 'Weighted_sum'):
            for mode in ('dense', 'lds'):
                path = os.path.join(d, '%s_%s.pt' % (structure, mode))
                env = dict(os.environ, EAGCN_AGG=mode)
                r = subprocess.run([sys.executable, '-c', code, structure, path], env=env, capture_output=True, text=True, timeout=300)
                assert r.returncode == 0, r.stderr[-2000:]
                res[(structure, mode)] = torch.load(path)
    for structure in ('Concate', 'Weighted_sum'):
        a, b = res[(structure, 'dense')], res[(structure, 'lds')]
        assert ((a['out'] - b['out']).abs().max() / a['out'].abs().max()).item() < 1e-5
        scale = max(v.abs().max().item() for v in a['g'].values())
        for k, v in a['g'].items():
            dd = (b['g'][k] - v).abs().max().item()
            assert dd <= 1e-5 * v.abs().max().item() + 2e-6 * scale, (structure, k, dd, v.abs().max().item(), scale)


def test_hidden_layers_without_fp32_twin_are_bit_identical():
    """Round 6: where the layer above reads its input from the plane images in both directions, a hidden layer stores no fp32 output
    (csrc/layer.hip layer_reads_planes_only, bn_apply).  EAGCN_PLANES_ONLY=0 keeps the fp32 twin: the same arithmetic either way, so
    outputs and every gradient must be BIT-identical -- eager engine and captured step, three layers (two hidden outputs), widths
    that take the plane path (>= 128 columns) and, as the control, widths that do not (the switch then changes nothing at all).
    Subprocesses: the switch is read once per process."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import sys, torch
sys.path.insert(0, %r)
from eagcn_amd import EAGCN
from eagcn_amd.synthetic import bce_weights, make_batch
w1, graph = int(sys.argv[1]), sys.argv[2] == 'graph'
mb = make_batch(B=40, n_max=50, n_med=18, rel_channels=(28, 4, 2, 2, 2), seed=33, n_tasks=3)
dense = [t.cuda() for t in mb.dense()]
labels = torch.from_numpy(mb.labels).cuda()
bw = torch.tensor(bce_weights(3), dtype=torch.float32, device='cuda')
torch.manual_seed(3)
m = EAGCN(28, 24, *[w1] * 5, *[w1 + 16] * 5, 64, 32, 3, 0.25, structure=sys.argv[3], n_layers=3, grad_mode='direct', graph=graph).cuda().train()
torch.manual_seed(11)                       # (the dropout seeds of the step come from the host generator)
loss, (out, _, gr) = m.fused_step(dense, labels, 'class', bw) if graph else (None, m(*dense))
if not graph:
    from eagcn_amd.losses import fused_classification_loss
    loss = fused_classification_loss(out, labels, bw)
    loss.backward()
torch.cuda.synchronize()
torch.save({'loss': loss.detach().cpu(), 'out': out.detach().cpu(), 'g': {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None}}, sys.argv[4])
''' % root
    with tempfile.TemporaryDirectory() as d:
        for structure in ('Concate', 'Weighted_sum'):
            # input columns of the layers above: Concate 5 x 48 = 240 / 320 (planes), 5 x 16 = 80 (none: the control);
            # Weighted_sum: the merged width itself, 128 / 144 (planes)
            for w1 in ((48, 16) if structure == 'Concate' else (128,)):
                for mode in ('eager', 'graph'):
                    res = {}
                    for sw in ('1', '0'):
                        path = os.path.join(d, 'r_%s_%d_%s_%s.pt' % (structure, w1, mode, sw))
                        env = dict(os.environ, EAGCN_PLANES_ONLY=sw)
                        r = subprocess.run([sys.executable, '-c', code, str(w1), mode, structure, path], env=env, capture_output=True,
                                           text=True, timeout=300)
                        assert r.returncode == 0, r.stderr[-2000:]
                        res[sw] = torch.load(path)
                    a, b = res['1'], res['0']
                    assert torch.equal(a['out'], b['out']) and torch.equal(a['loss'], b['loss']), (structure, w1, mode)
                    assert a['g'].keys() == b['g'].keys()
                    for k in a['g']:
                        assert torch.equal(a['g'][k], b['g'][k]), (structure, w1, mode, k)
