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
    assert lib.eagcn_abi_version() == 1
    assert lib.eagcn_pad16(140) == 144 and lib.eagcn_pad16(80) == 80


def test_ctypes_structs_match_c_layout():
    import ctypes as C
    from eagcn_amd import _lib
    lib = _lib.load()
    for which, cls in enumerate((_lib.Batch, _lib.Layout, _lib.LayerParams, _lib.LayerBufs, _lib.LayerGrads,
                                 _lib.HeadParams, _lib.HeadGrads, _lib.Model)):
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


@pytest.mark.parametrize('name', [n for n in golden_cases('model') if n in ('model_concate_train', 'model_weighted_train')])
def test_module_tree_equals_reference_state_dict(name):
    from eagcn_amd import EAGCN
    g = Golden(name)
    m = g.meta
    model = EAGCN(m['n_bfeat'], m['n_afeat'], *m['widths1'], *m['widths2'], m['dens'][0], m['dens'][1],
                  m['nclass'], 0.0, structure=m['structure'], molfp_mode=m['molfp'])
    sd = g.state_dict()
    model.load_state_dict(sd, strict=True)
    assert list(model.state_dict().keys()) == list(sd.keys())        # same keys, same order
    assert len(sd) == 246 if m['structure'] == 'Concate' else len(sd) > 246
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
    with pytest.raises(ValueError):
        EAGCN(28, 24, *[8] * 5, *[8] * 5, 8, 8, 1, 0.0, structure='GAT')       # baseline archs are out of scope


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
