"""molfp_mode='pool' -- the Diff_Pooling read-out (reference layers.py:492-506, models.py:90-92, 104-106) on csrc/pool.hip.

The five reference fixtures (tests/golden/model_*_pool_*.npz) run through test_gpu_parity.py::test_model_golden and
test_compact_input_golden like every other model fixture (F = 24..64 columns, N = 12).  Here: the pieces against a float64
tensor-op restatement (so that a failure names the kernel), and whole models against the CPU oracle at shapes that take
several column chunks (F > 64), several lane trips per attention row (N > 64) and other cluster counts."""
import numpy as np
import pytest
import torch

from helpers import assert_grad_close, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _nat(adj):
    """Stored rows per molecule as the batch index counts them: last row with a bond + 1."""
    has = adj.max(dim=2)[0] > 0
    idx = torch.arange(adj.shape[1]).view(1, -1) + 1
    return (has * idx).max(dim=1)[0]


def _unpack(packed, nat, N):
    """Packed rows (molecule-major, nat[b] rows each) -> zero-padded [B, N, width] on the host."""
    rows = packed.detach().double().cpu()
    out = torch.zeros(len(nat), N, rows.shape[1], dtype=torch.float64)
    r = 0
    for b, n in enumerate(int(v) for v in nat):
        out[b, :n] = rows[r:r + n]
        r += n
    return out


def _ref_attention(adj, rels, att, ave_a, self_r, mode):
    B, N, _ = adj.shape
    m = adj.max(dim=2, keepdim=True)[0]
    eye = torch.eye(N, dtype=adj.dtype)
    if mode == 'gat':                                                       # layers.py:189
        return adj + m * eye
    if mode == 'gcn':                                                       # layers.py:250-253
        u = adj + m * eye + (1.0 - adj) * 1e-9
    else:                                                                   # layers.py:82-83, 318-322
        a1 = [torch.sigmoid((r * w.view(1, -1, 1, 1)).sum(1)) * adj for r, w in zip(rels, att)]
        a = sum(wa * a1k for wa, a1k in zip(ave_a, a1))
        u = torch.sigmoid(a) * adj + torch.sigmoid(self_r) * (m * eye) + (1.0 - adj) * 1e-9
    return (u / u.sum(dim=2, keepdim=True)) * m


def _ref_pool(a, x, wf, ws):
    B, N, _ = a.shape
    flat = torch.bmm(a, x).reshape(B * N, -1)
    xf = torch.relu(flat @ wf).view(B, N, -1)
    s = torch.softmax((flat @ ws).view(B, N, -1), dim=2)
    return torch.relu(torch.bmm(s.transpose(1, 2), xf)).sum(1)               # layers.py:499-503, models.py:106


@pytest.mark.parametrize('mode,with_pad,F,P,n_max', [('attention', False, 40, 5, 30), ('attention', True, 150, 5, 70),
                                                     ('gcn', True, 70, 3, 20), ('gat', False, 33, 8, 90),
                                                     ('attention', True, 24, 1, 12)])
def test_pool_readout_pieces_against_float64_tensor_ops(mode, with_pad, F, P, n_max):
    from eagcn_amd import ops
    from eagcn_amd.synthetic import make_batch
    g = torch.Generator().manual_seed(21)
    channels = (6, 4, 2, 2, 2)
    mb = make_batch(B=6, n_max=n_max, n_med=max(4, n_max // 3), rel_channels=channels, seed=4, isolated_frac=0.15)
    dense = mb.dense()
    adj, rels = dense[0], list(dense[2:-1])
    if mode != 'attention':
        rels = rels[:1]
    B, N, _ = adj.shape
    nat = _nat(adj)
    stored = (torch.arange(N).view(1, -1) < nat.view(-1, 1)).view(B, N, 1)
    att = [torch.randn(c, generator=g) for c in channels[:len(rels)]]
    ave_a = torch.rand(len(rels), generator=g) * 1.2 - 0.6
    self_r = torch.rand(1, generator=g) - 0.5
    x_free = torch.randn(B, N, F, generator=g)
    pad_row = torch.randn(F, generator=g) * 0.5 if with_pad else None
    wf = torch.randn(F, F, generator=g) * 0.2
    ws = torch.randn(F, P, generator=g) * 0.3
    gsel = torch.randn(B, F, generator=g)

    # ---- float64 restatement -----------------------------------------------------------------------------------------------
    leaves = [t.double().requires_grad_(True) for t in att] + [t.double().requires_grad_(True) for t in (ave_a, self_r, x_free, wf, ws)]
    att64, (ave64, r64, x64, wf64, ws64) = leaves[:len(att)], leaves[len(att):]
    pad64 = pad_row.double().requires_grad_(True) if with_pad else None
    a64 = _ref_attention(adj.double(), [r.double() for r in rels], att64, ave64, r64, mode)
    fill = pad64.view(1, 1, F) if with_pad else torch.zeros(1, 1, F, dtype=torch.float64)
    xfull = torch.where(stored, x64, fill.expand(B, N, F))
    g64 = _ref_pool(a64, xfull, wf64, ws64)
    (g64 * gsel.double()).sum().backward()

    # ---- HIP ---------------------------------------------------------------------------------------------------------------
    dev = torch.device('cuda', 0)
    index = ops.BatchIndex(adj.to(dev), [r.to(dev) for r in rels], bond_lists=(mode == 'gat'))
    layout = ops.ColLayout.single(F, 16)
    attd = [t.view(1, -1, 1, 1).to(dev).requires_grad_(True) for t in att]
    aved, rd = ave_a.to(dev).requires_grad_(True), self_r.to(dev).requires_grad_(True)
    xd = x_free.to(dev).requires_grad_(True)
    padd = None
    if with_pad:
        padd = torch.zeros(layout.ld, device=dev)
        padd[:F] = pad_row.to(dev)
        padd.requires_grad_(True)
    wfd, wsd = wf.to(dev).requires_grad_(True), ws.to(dev).requires_grad_(True)
    xp = ops.pack_rows(index, layout, xd)
    kw = dict(att_w=attd, ave_a=aved, self_r=rd) if mode == 'attention' else {}
    # the attention matrix on its own
    A, _rinv, padsum = ops._PoolAttention.apply(index, ops.POOL_MODES[mode], kw.get('ave_a'), kw.get('self_r'), *kw.get('att_w', ()))
    a_dense = _unpack(A, nat, N)[:, :, :N]
    a_ref = a64.detach() * stored.double()                                  # rows beyond nat have no bond: zero anyway
    assert rel_err(a_dense, a_ref, 'A') < 2e-6
    tail_ref = torch.stack([a64[b, i, int(nat[b]):].sum() for b in range(B) for i in range(int(nat[b]))]).detach()
    assert (padsum[:tail_ref.numel()].double().cpu() - tail_ref).abs().max().item() <= 1e-6 * max(tail_ref.abs().max().item(), 1e-12)
    # A.x
    AX = ops._PoolMix.apply(index, layout, A, padsum, xp, padd)
    ax_dense = _unpack(AX, nat, N)
    ax_ref = torch.bmm(a64, xfull).detach() * stored.double()
    assert rel_err(ax_dense, ax_ref, 'A.x') < 2e-6
    # the whole read-out and every gradient
    gh = ops.pool_readout(index, layout, xp, padd, wfd, wsd, mode=mode, **kw)
    assert rel_err(gh.detach().cpu(), g64.detach(), 'g') < TOL
    (gh * gsel.to(dev)).sum().backward()
    got = {'wf': wfd.grad, 'ws': wsd.grad, 'x': xd.grad}
    want = {'wf': wf64.grad, 'ws': ws64.grad, 'x': x64.grad * stored.double()}
    if with_pad:
        got['pad_row'], want['pad_row'] = padd.grad[:F], pad64.grad
        assert float(padd.grad[F:].abs().max()) == 0.0 if layout.ld > F else True
    if mode == 'attention':
        got['ave_a'], want['ave_a'] = aved.grad, ave64.grad
        got['self_r'], want['self_r'] = rd.grad, r64.grad
        for k in range(len(att)):
            got['att%d' % k], want['att%d' % k] = attd[k].grad.view(-1), att64[k].grad
    # Pm = S^T relu(.) is never negative and the rows of S sum to one, so the cluster sum of models.py:106 does not depend
    # on S at all: d ws is analytically zero (summation noise in both implementations), hence the common floor.  d pad_row
    # only exists through the 1e-9 filler weights: against its own magnitude.
    scale = max(v.abs().max().item() for k, v in want.items() if k != 'pad_row')
    for k in want:
        assert_grad_close(got[k], want[k], want[k].abs().max().item() if k == 'pad_row' else scale, k, rtol=2e-5, floor=2e-6)


def _models(structure, w1, w2, nclass, pool_num, training):
    from eagcn_amd import EAGCN
    from oracle.eagcn_ref import RefEAGCN, weights_init_
    torch.manual_seed(8)
    ref = RefEAGCN(28, 24, w1, w2, 64, 32, nclass, 0.0, structure=structure, molfp_mode='pool', pool_num=pool_num)
    weights_init_(ref)
    with torch.no_grad():                                                   # weights_init leaves the pooling bases at N(0, .02):
        for p in (ref.pool1.feature_layer.weight, ref.pool1.adjacent_layer.weight):    # widen them so that relu / softmax matter
            p.mul_(8.0)
    hip = EAGCN(28, 24, *w1, *w2, 64, 32, nclass, 0.0, structure=structure, molfp_mode='pool', pool_num=pool_num).cuda()
    hip.load_state_dict(ref.state_dict(), strict=True)
    ref.train(training)
    hip.train(training)
    return ref, hip


@pytest.mark.parametrize('structure,pool_num,compact', [('Concate', 5, False), ('Weighted_sum', 5, True), ('GCN', 3, False),
                                                        ('GAT', 8, False), ('Concate', 2, True)])
def test_pool_model_vs_oracle_wide(structure, pool_num, compact):
    """f_last = 192 columns (three 64-column chunks), N = 70 (two lane trips per attention row), isolated atoms, other
    cluster counts than 5; training mode except for GAT (its attention dropout is not configurable: layers.py:104)."""
    from eagcn_amd.synthetic import make_batch
    w1, w2 = [20, 16, 12, 10, 14], [30, 20, 16, 12, 18]
    training = structure != 'GAT'
    import copy
    ref, hip = _models(structure, w1, w2, 4, pool_num, training)
    ref64 = copy.deepcopy(ref).double()                                     # float64 oracle: arbiter for ill-conditioned entries
    mb = make_batch(B=7, n_max=70, n_med=20, rel_channels=(28, 4, 2, 2, 2), seed=12, isolated_frac=0.1)
    cpu = mb.dense()
    gsel = torch.randn(7, 4)
    if compact:
        bonds, afm, size = mb.compact('cuda')
        out_h, rep_h, gr_h = hip.forward_compact(bonds, afm, size)
    else:
        out_h, rep_h, gr_h = hip(*[t.cuda() for t in cpu])
    (out_h * gsel.cuda()).sum().backward()
    gh = {k: p.grad for k, p in hip.named_parameters() if p.grad is not None}
    cand = []
    for m, inp in ((ref, cpu), (ref64, [t.double() if t.is_floating_point() else t for t in cpu])):
        out_r, rep_r, gr_r = m(*inp)
        (out_r * gsel.to(out_r.dtype)).sum().backward()
        cand.append((out_r.detach(), gr_r.detach(), rep_r, {k: p.grad for k, p in m.named_parameters() if p.grad is not None}))
    assert min(rel_err(out_h.detach().cpu(), c[0], 'out') for c in cand) < TOL
    assert min(rel_err(gr_h.detach().cpu(), c[1], 'graph_rep') for c in cand) < TOL
    assert rel_err(rep_h.cpu(), cand[0][2], 'atom_rep') < TOL
    assert set(cand[0][3]) == set(gh), set(cand[0][3]) ^ set(gh)
    scale = max(v.abs().max().item() for v in cand[0][3].values())
    for k in gh:
        try:
            assert_grad_close(gh[k], cand[0][3][k], scale, k, rtol=2e-5, floor=2e-6)
        except AssertionError:
            e_ref = (cand[0][3][k].double() - cand[1][3][k]).abs().max().item()
            e_hip = (gh[k].double().cpu() - cand[1][3][k]).abs().max().item()
            assert e_hip <= 4.0 * e_ref + 2e-6 * scale, (k, e_hip, e_ref)
    if training:                                                            # running statistics of every BatchNorm advanced once
        sd_h, sd_r = hip.state_dict(), ref.state_dict()
        for k in sd_r:
            if 'running' in k or 'num_batches' in k:                    # (bn_den1 follows a BatchNorm: its mean is summation noise)
                d = (sd_h[k].double().cpu() - sd_r[k].double()).abs().max().item()
                assert d <= TOL * sd_r[k].double().abs().max().item() + 1e-6, (k, d)
