"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol that
include/eagcn_hip.h declares (no compute calls without a GPU), the ctypes structs agree with the C
structs, the module tree equals the reference's state_dict, the product path refuses CPU tensors
and a missing library instead of falling back, the synthetic batches obey the collate contract."""
import os
import re

import numpy as np
import pytest
import torch

from helpers import Golden, golden_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'eagcn_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(eagcn_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()                                       # hipcc cross-compiles gfx950 without a GPU
    from eagcn_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), 'libeagcn_hip.so does not export %s' % name
    assert set(declared) == set(_lib.SIGNATURES), set(declared) ^ set(_lib.SIGNATURES)
    assert lib.eagcn_abi_version() == _lib.ABI_VERSION == 7
    assert lib.eagcn_pad16(140) == 144 and lib.eagcn_pad16(80) == 80


def test_ctypes_structs_match_c_layout():
    import ctypes as C
    from eagcn_amd import _lib
    lib = _lib.load()
    for which, cls in enumerate((_lib.Batch, _lib.Layout, _lib.LayerParams, _lib.LayerBufs, _lib.LayerGrads,
                                 _lib.HeadParams, _lib.HeadGrads, _lib.Model, _lib.GatParams, _lib.PoolAtt)):
        assert lib.eagcn_struct_size(which) == C.sizeof(cls), cls.__name__


def test_argument_errors_are_reported_not_crashed():
    import ctypes as C
    from eagcn_amd import _lib
    lib = _lib.load()
    b = _lib.Batch()
    rc = lib.eagcn_index_build(None, None, C.byref(b), None, None)       # null arguments, no GPU needed
    assert rc == -1
    assert b'null' in lib.eagcn_last_error()
    with pytest.raises(_lib.EagcnHipError):
        _lib.check(rc, 'eagcn_index_build')


@pytest.mark.parametrize('name', [n for n in golden_cases('model') if n in ('model_concate_train', 'model_weighted_train', 'model_concate_pool_train')])
def test_module_tree_equals_reference_state_dict(name):
    from eagcn_amd import EAGCN
    g = Golden(name)
    m = g.meta
    model = EAGCN(m['n_bfeat'], m['n_afeat'], *m['widths1'], *m['widths2'], m['dens'][0], m['dens'][1],
                  m['nclass'], 0.0, structure=m['structure'], molfp_mode=m['molfp'])
    sd = g.state_dict()
    model.load_state_dict(sd, strict=True)
    assert list(model.state_dict().keys()) == list(sd.keys())        # same keys, same order
    assert len(sd) == 246 + (4 if m['molfp'] == 'pool' else 0) if m['structure'] == 'Concate' else len(sd) > 246
    # attribute walk of check_model.py:48-58
    assert model.layer1.block1.att.weight.shape == (1, m['n_bfeat'], 1, 1)
    assert model.layer4.block5.batch_norm.bn.running_mean.shape[0] == model.layer4.widths[4]


def test_aliases_and_layer_count_extension():
    from eagcn_amd import Concate_GCN, Weighted_GCN
    m2 = Concate_GCN(28, 24, *[80] * 5, *[140] * 5, 256, 64, 12, 0.3, n_layers=2)
    assert m2.structure == 'Concate' and not hasattr(m2, 'layer3') and m2.den1.in_features == 700
    m3 = Weighted_GCN(28, 24, *[10] * 5, *[20] * 5, 32, 16, 1, 0.0, n_layers=3)
    assert m3.structure == 'Weighted_sum' and m3.den1.in_features == 200 and m3.layer3.widths == [200] * 5
    from eagcn_amd import EAGCN
    gat = EAGCN(28, 24, *[8] * 5, *[8] * 5, 8, 8, 1, 0.0, structure='GAT')     # baselines of models.py:63-73
    assert gat.layer4.graph_conv.W.shape == (40, 80) and gat.layer1.graph_conv.a.shape == (80, 1) and gat.den1.in_features == 80
    g2 = EAGCN(28, 24, *[8] * 5, *[8] * 5, 8, 8, 1, 0.0, structure='GAT', graph=True)   # captured layer-by-layer step
    assert g2.graph and g2.structure == 'GAT'                                          # (graph_composed.ComposedRunner)
    with pytest.raises(ValueError):
        EAGCN(28, 24, *[8] * 5, *[8] * 5, 8, 8, 1, 0.0, structure='Pool')


def test_no_cpu_fallback():
    from eagcn_amd import EAGCN, _lib
    from eagcn_amd.synthetic import make_batch
    mb = make_batch(B=2, n_max=6, n_med=4, rel_channels=(4, 4, 2, 2, 2), seed=0)
    model = EAGCN(4, 24, *[4] * 5, *[4] * 5, 8, 4, 1, 0.0, n_layers=2)
    with pytest.raises(_lib.EagcnHipError, match='no CPU'):
        model(*mb.dense())


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from eagcn_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'libeagcn_hip.so'))
    with pytest.raises(_lib.EagcnHipError, match='not built'):
        _lib.load()


def test_synthetic_batch_obeys_collate_contract():
    from eagcn_amd.synthetic import make_batch
    mb = make_batch(B=9, n_max=30, n_med=10, rel_channels=(7, 4, 2, 2, 2), seed=3, isolated_frac=0.1)
    adj, afm, *rels, size = mb.dense()
    assert adj.shape == (9, 30, 30) and afm.shape == (9, 30, 24) and size.dtype == torch.int64
    assert torch.equal(adj, adj.transpose(1, 2)) and adj.diagonal(dim1=1, dim2=2).abs().sum() == 0
    assert set(adj.unique().tolist()) <= {0.0, 1.0}
    for r, c in zip(rels, (7, 4, 2, 2, 2)):
        assert r.shape == (9, c, 30, 30)
        assert torch.equal(r.sum(1), adj)                      # exactly one hot channel on every bond
        assert torch.equal(r, r.transpose(2, 3))
    for b in range(9):
        n = int(size[b])
        assert afm[b, n:].abs().sum() == 0 and adj[b, n:].abs().sum() == 0
    assert int(size.max()) == 30


def test_losses_match_oracle():
    from eagcn_amd import losses
    from oracle import eagcn_ref
    torch.manual_seed(0)
    out = torch.randn(7, 5)
    labels = torch.from_numpy(np.random.default_rng(0).choice([0.0, 1.0, -1.0], size=(7, 5)).astype(np.float32))
    bw = [[3.0 + j, 0.5 + 0.1 * j] for j in range(5)]
    a = losses.classification_loss(out, labels, torch.tensor(bw))
    b = eagcn_ref.classification_loss(out, labels, bw)
    assert abs(float(a) - float(b)) < 1e-6
    assert abs(float(losses.regression_loss(out[:, :1], labels[:, :1])) -
               float(eagcn_ref.regression_loss(out[:, :1], labels[:, :1]))) < 1e-7


def test_algorithmic_flops_match_survey_table():
    import bench
    cfg = bench.WORKLOADS['tox21_c2']
    per_mol = bench.algorithmic_flops(cfg, [132]) / 1e6
    assert abs(per_mol - 315) < 5, per_mol                      # 315 MF at n = 132
    assert abs(bench.algorithmic_flops(cfg, [53]) / 1e6 - 106) < 3


def test_compat_modules_expose_reference_names():
    import importlib
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'compat'))
    try:
        for n in ('models', 'layers'):
            sys.modules.pop(n, None)
        models = importlib.import_module('models')
        layers = importlib.import_module('layers')
        assert models.EAGCN.__module__ == 'eagcn_amd.models'
        for name in ('GraphConv_Layer', 'GraphConv_block', 'GraphConv_base', 'AFM_BatchNorm', 'Ave_multi_view', 'Dense'):
            assert hasattr(layers, name)
    finally:
        sys.path.remove(os.path.join(ROOT, 'compat'))
        sys.modules.pop('models', None)
        sys.modules.pop('layers', None)


def test_eval_metrics_match_sklearn():
    """train.py:156-186 (roc_curve + auc per task over the labelled entries) and train.py:188-211 (RMSE) as tensor ops."""
    from sklearn import metrics
    from eagcn_amd.training import auc_per_task, rmse, set_weight
    g = torch.Generator().manual_seed(3)
    n, T = 500, 4
    scores = torch.rand(n, T, generator=g)
    scores[:, 1] = (scores[:, 1] * 5).round() / 5                   # heavy ties
    labels = torch.randint(-1, 2, (n, T), generator=g).float()      # -1 = missing
    labels[:, 3] = torch.where(labels[:, 3] == 1, torch.zeros(()), labels[:, 3])   # a task without positives
    valid = (labels == 0) | (labels == 1)
    aucs, mean = auc_per_task(scores, labels, valid)
    want = []
    for j in range(T):
        m = valid[:, j].numpy()
        y, s = labels[:, j].numpy()[m], scores[:, j].numpy()[m]
        if y.min() == y.max():
            want.append(float('nan'))
            continue
        fpr, tpr, _ = metrics.roc_curve(y.astype(int), s, pos_label=1)
        want.append(metrics.auc(fpr, tpr))
    for a, w in zip(aucs, want):
        assert (np.isnan(a) and np.isnan(w)) or abs(a - w) < 1e-12, (aucs, want)
    assert abs(mean - np.nanmean(want)) < 1e-12
    pred, tgt = torch.randn(300, 1, generator=g), torch.randn(300, 1, generator=g)
    assert abs(rmse(pred, tgt) - np.sqrt(metrics.mean_squared_error(pred.numpy().ravel(), tgt.numpy().ravel()))) < 1e-6
    with pytest.raises(KeyError):                  # task 3 has no positives: the reference has no weight for it either
        set_weight(labels, T)


def test_set_weight_and_weight_tensor_match_the_reference(golden_dir):
    """utils.py:681-700 / 653-679, lifted out of the reference by tools/make_collate_golden.py (inputs + outputs only)."""
    from eagcn_amd.losses import class_weight_tensor
    from eagcn_amd.training import set_weight
    z = np.load(os.path.join(golden_dir, 'class_weights.npz'))
    w = set_weight(z['y_all'])
    assert np.array_equal(np.array(w, dtype=np.float64), z['weights'])          # 5000 / count, bit for bit
    assert w == set_weight(torch.from_numpy(z['y_all']), z['y_all'].shape[1])
    wt = class_weight_tensor(w, torch.from_numpy(z['batch']))
    assert np.array_equal(wt.numpy(), z['weight_tensor'])


def test_model_pickles_and_deepcopies_after_planning():
    """The reference checkpoints with torch.save(model, ...) (train.py:440): the cached ctypes descriptors / graph runners
    must not travel with the module."""
    import copy
    import io
    from eagcn_amd import EAGCN
    m = EAGCN(7, 24, *[8, 6, 4, 4, 5], *[10, 7, 5, 6, 4], 16, 8, 3, 0.2, n_layers=2, graph=True)
    plan = m.plan()
    plan.cmodel(True, 1, 0.2)                       # populates the ctypes cache (no GPU needed)
    m._runners['fake'] = object()
    c = copy.deepcopy(m)
    assert c._plan is None and c._runners == {} and m._plan is plan
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    r = torch.load(buf, weights_only=False)
    assert r._plan is None and r._runners == {}
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), r.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_pool_constructor_contract():
    from eagcn_amd import EAGCN
    assert EAGCN(28, 24, *[8] * 5, *[8] * 5, 8, 8, 1, 0.0, molfp_mode='pool', graph=True).graph   # captured layer-by-layer step
    with pytest.raises(ValueError):
        EAGCN(28, 24, *[8] * 5, *[8] * 5, 8, 8, 1, 0.0, molfp_mode='pool', n_layers=2)       # A of layers.py:319-324 needs layer 4
    with pytest.raises(ValueError):
        EAGCN(28, 24, *[8] * 5, *[8] * 5, 8, 8, 1, 0.0, molfp_mode='pool', pool_num=9)
    m = EAGCN(28, 24, *[8] * 5, *[8] * 5, 8, 8, 1, 0.0, structure='GCN', molfp_mode='pool', n_layers=2)
    assert m.pool1.adjacent_layer.weight.shape == (40, 5) and m.pool3.adjacent_layer.weight.shape == (40, 1)


def test_flat_gradient_layout_is_16_byte_aligned_per_parameter():
    """Every parameter's slot of the flat gradient buffer starts on a 16-byte boundary (the GEMM epilogue that writes
    blockK.graph_conv.weight.grad uses 16-byte stores; a 1-element self_r sits right in front of that matrix) and the views
    have the parameters' shapes, in parameter order."""
    from eagcn_amd import EAGCN
    m = EAGCN(28, 24, *[8] * 5, *[12] * 5, 16, 8, 3, 0.0, n_layers=2)
    p = m.plan()
    assert all(o % 4 == 0 for o in p.offsets)
    assert p.offsets[-1] >= sum(p.sizes) and p.offsets[-1] - sum(p.sizes) < 4 * len(p.sizes)
    flat = torch.arange(p.offsets[-1], dtype=torch.float32)
    views = p.grad_views(flat)
    assert [tuple(v.shape) for v in views] == [tuple(q.shape) for q in p.params]
    for o, v in zip(p.offsets, views):
        assert v.data_ptr() == flat.data_ptr() + 4 * o
    seen = torch.zeros(p.offsets[-1], dtype=torch.bool)
    for o, n in zip(p.offsets, p.sizes):
        assert not seen[o:o + n].any()
        seen[o:o + n] = True


def test_xcd_local_gemm_schedule_partitions_the_iteration_space():
    """csrc/kernels.h g3_plan (the XCD-local schedule of the paired backward products): the eight segments tile the row blocks
    of dX and the k-steps of dW exactly once, in order, and each segment's share of the work follows its share of the waves."""
    import ctypes as C
    from eagcn_amd import _lib
    lib = _lib.load()
    cases = [(4809, 400, 704, 400, 704, 4809, 256), (19200, 400, 704, 400, 704, 19200, 256), (25000, 2500, 6320, 2500, 6320, 25000, 256),
             (100, 128, 128, 128, 128, 100, 256), (64, 400, 704, 400, 704, 64, 256), (4809, 400, 704, 400, 704, 4809, 100),
             (4809, 400, 704, 400, 704, 4809, 8), (1, 128, 16, 128, 16, 1, 256), (70000, 128, 144, 128, 144, 70000, 304)]
    for M0, N0, K0, M1, N1, K1, G in cases:
        a, c = (C.c_int * 9)(), (C.c_int * 9)()
        assert lib.eagcn_gemm_sk_plan(M0, N0, K0, M1, N1, K1, G, a, c) == 0
        a, c = list(a), list(c)
        RB, CT0, ipt0 = -(-M0 // 64), -(-N0 // 64), max(1, -(-K0 // 16))
        T1, ipt1 = -(-M1 // 64) * -(-N1 // 64), max(1, -(-K1 // 16))
        assert a[0] == 0 and a[8] == RB and c[0] == 0 and c[8] == ipt1, (a, c)
        assert all(a[i] <= a[i + 1] for i in range(8)) and all(c[i] <= c[i + 1] for i in range(8)), (a, c)
        total = RB * CT0 * ipt0 + T1 * ipt1
        if total > 64 * G:       # a segment starts within half a row of k-steps (T1 / 2 iterations) of its XCD's first wave range
            R, qn, rn = 4 * G, G // 8, G % 8
            per, rem = divmod(total, R)
            for x in range(9):
                L = 4 * (x * qn + min(x, rn))
                assert abs(a[x] * CT0 * ipt0 + T1 * c[x] - (L * per + min(L, rem))) <= T1 // 2 + 1, (x, a, c)


def test_plane_image_layout_and_split_policy_host_side():
    """The operand format of the plane GEMM (include/eagcn_hip.h, csrc/bx3.h): the documented element -> index map is a bijection
    onto the image for every (rows, ld) and the image size is what eagcn_bx3_plane_elems returns; the k-chunk policy of the weight
    gradient (pure host functions of the C ABI: no GPU involved) keeps its promises -- between 1 and the slab capacity, chunks of
    at most 4096 rows whenever the capacity allows it, at least 8 k-tiles per chunk beside a dX product, 24 alone."""
    from eagcn_amd import _lib
    lib = _lib.load()
    for rows, ld in ((1, 8), (37, 64), (100, 400), (4809, 720), (5, 40)):
        n = int(lib.eagcn_bx3_plane_elems(rows, ld))
        assert n == (ld + 31) // 32 * 32 * rows
        r = np.arange(rows)[:, None]
        c = np.arange(ld)[None, :]
        idx = ((c >> 5) * rows + r) * 32 + ((((c >> 3) & 3) ^ ((r >> 2) & 3)) << 3) + (c & 7)
        assert idx.min() >= 0 and idx.max() < n and np.unique(idx).size == rows * ld
        # four adjacent columns (c a multiple of 4) are contiguous: the producers' 8-byte stores
        assert (np.diff(idx.reshape(rows, ld // 4, 4), axis=2) == 1).all()
    for (M, N, K, cap) in ((400, 720, 4809, 19), (400, 720, 19200, 64), (512, 6320, 25000, 64), (512, 1024, 262144, 64), (128, 128, 37, 4),
                           (400, 720, 100, 8)):
        alone = lib.eagcn_bx3_used_splits(cap, M, N, K)
        pair = lib.eagcn_bx3_pair_used_splits(cap, M, N, K, K, M, N)
        kt = max(1, -(-K // 32))
        for s, min_kt in ((alone, 24), (pair, 8)):
            assert 1 <= s <= cap
            if cap >= -(-K // 4096):
                assert -(-kt // s) * 32 <= 4096 + 31, (M, N, K, s)
            assert s == 1 or s == -(-K // 4096) or kt // s >= min_kt - 1 or kt // min_kt < 1, (M, N, K, s, min_kt)


def test_pad_row_binomial_thresholds_are_the_binomial_cdf():
    """tests/helpers.py pad_thresholds = the numpy mirror of csrc/readout.hip pad_binomial_table_kernel (the kept-row counts of the
    non-stored rows of a Weighted_sum layer under dropout are drawn by inverting these thresholds; the GPU test with injected masks
    pins kernel == mirror): the mirror IS the binomial distribution -- every threshold within one unit of 2^-32 of scipy's CDF,
    non-decreasing, the last one saturated."""
    from scipy.stats import binom
    from helpers import pad_thresholds
    for m, q in ((0, 0.7), (1, 0.7), (5, 0.7), (60, 0.5), (197, 0.7), (222, 0.9), (1000, 0.7), (1024, 0.3)):
        t = pad_thresholds(m, np.float64(q)).astype(np.float64)
        ref = np.floor(binom.cdf(np.arange(m + 1), m, q) * 2.0 ** 32)
        assert np.abs(t - np.minimum(ref, 4294967295.0)).max() <= 1.0, (m, q)
        assert (np.diff(t) >= 0).all() and t[-1] == 4294967295.0


def test_bond_list_policy_by_shape_and_structure():
    """Which batch shapes carry bond lists / row blocks for the LDS-staged aggregation (csrc/lagg.hip lagg_wanted; no kernel runs):
    large padded sizes, batches of up to 256 molecules, every Concate shape and -- since round 6, when the transposed kernel re-forms
    dH of a Weighted_sum layer from the upstream gradient in its staging -- Weighted_sum layers of small molecules in large batches
    too (EAGCN_LAGG_WFUSE=0 with the round-5 forward limit EAGCN_LAGG_FWD_MAXB=256: those stay on the matrix-core kernels); nothing beyond 256 atoms.  In a subprocess: the policy reads
    its environment once."""
    import subprocess
    import sys
    code = r'''
import os
for k in ('EAGCN_AGG', 'EAGCN_LAGG_MIN_N', 'EAGCN_LAGG_FWD_MAXB', 'EAGCN_LAGG_WFUSE'):
    os.environ.pop(k, None)
from eagcn_amd import _lib as L
lib = L.load()
f = lib.eagcn_agg_wants_bond_lists_for
C, W = L.STRUCT_CONCATE, L.STRUCT_WEIGHTED
got = [f(256, 132, C), f(1024, 132, C), f(1024, 222, W), f(256, 222, W), f(1024, 256, W), f(64, 300, C), f(1024, 222, -1),
       lib.eagcn_agg_wants_bond_lists(1024, 222), lib.eagcn_agg_wants_bond_lists(8, 257)]
print(got)
assert got == [1, 1, 1, 1, 1, 0, 1, 1, 0], got
'''
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout + r.stderr
    forced = subprocess.run([sys.executable, '-c', code.replace("os.environ.pop(k, None)", "os.environ.pop(k, None)\nos.environ['EAGCN_AGG'] = 'dense'")
                             .replace("assert got == [1, 1, 1, 1, 1, 0, 1, 1, 0], got", "assert got == [0] * 9, got")],
                            capture_output=True, text=True, cwd=str(ROOT))
    assert forced.returncode == 0, forced.stdout + forced.stderr
    off = subprocess.run([sys.executable, '-c', code.replace("os.environ.pop(k, None)", "os.environ.pop(k, None)\nos.environ['EAGCN_LAGG_WFUSE'] = '0'\nos.environ['EAGCN_LAGG_FWD_MAXB'] = '256'")
                          .replace("assert got == [1, 1, 1, 1, 1, 0, 1, 1, 0], got", "assert got == [1, 1, 0, 1, 1, 0, 1, 1, 0], got")],
                         capture_output=True, text=True, cwd=str(ROOT))
    assert off.returncode == 0, off.stdout + off.stderr


def test_bond_list_buffers_are_laid_out_without_overlap():
    """_lib.set_bond_lists: row / column list headers, per-molecule records and the row-block records (two int4 per block, at most one
    block per molecule, 16-byte aligned) inside ONE int32 buffer of bond_ptrs_len(T, B) words; list entries with their 64-bit codes
    8-byte aligned inside the edge buffer."""
    from eagcn_amd import _lib as L
    for T, B, E in ((4809, 256, 10331), (7, 3, 5), (262144, 1024, 1100000), (1, 1, 1)):
        ptrs = torch.zeros(L.bond_ptrs_len(T, B) + 3, dtype=torch.int32)[3:]          # (a deliberately 4-byte aligned start)
        edges = torch.zeros(6 * E + 2, dtype=torch.int32)
        c = L.Batch()
        c.T, c.B = T, B
        L.set_bond_lists(c, 4096, ptrs, edges, E)
        lo, hi = ptrs.data_ptr(), ptrs.data_ptr() + 4 * ptrs.numel()
        assert c.row_ptr == lo and c.col_ptr == lo + 8 * T and c.mol_info == lo + 16 * T
        assert c.blk % 16 == 0 and c.blk >= c.mol_info + 16 * B
        assert c.blk + 32 * B <= hi                        # B blocks of two int4 records
        elo, ehi = edges.data_ptr(), edges.data_ptr() + 4 * edges.numel()
        assert c.nbr == elo and c.tnbr == elo + 4 * E
        assert c.ecode % 8 == 0 and c.ecode >= c.tnbr + 4 * E and c.tcode == c.ecode + 8 * E and c.tcode + 8 * E <= ehi
        assert c.ecnt == 4096 and c.edge0 == 4096 + 4 * B
