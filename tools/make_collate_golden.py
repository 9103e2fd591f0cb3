#!/usr/bin/env python3
"""Golden fixture for the compact collate: per-molecule inputs as the reference's dataset yields them and
the padded batch tensors its OWN collate functions build from them.

The reference's utils.py cannot be imported here (it pulls rdkit at module level), so the two collate functions are lifted out
of /root/reference/eagcn_pytorch/utils.py with ``ast`` at generation time, executed with numpy / torch and ``use_cuda = False``,
and only their INPUTS and OUTPUTS are stored (tests/golden/collate_*.npz).  Run in the build container:
    python tools/make_collate_golden.py"""
import ast
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eagcn_amd.synthetic import make_batch   # noqa: E402

REF = '/root/reference/eagcn_pytorch/utils.py'
OUT = os.path.join(ROOT, 'tests', 'golden')


def reference_collates():
    tree = ast.parse(open(REF).read())
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ('mol_collate_func_class', 'mol_collate_func_reg')]
    assert len(wanted) == 2
    ns = {'np': np, 'torch': torch, 'use_cuda': False, 'FloatTensor': torch.FloatTensor}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), REF, 'exec'), ns)
    return ns['mol_collate_func_class'], ns['mol_collate_func_reg']


def molecules(mb):
    """Per-molecule tuples (adj, afm, TypeAtt, OrderAtt, AromAtt, ConjAtt, RingAtt, label, smile, subtype, index) of a
    synthetic batch, each at its own size n (what MolDatum / the dataset __getitem__ deliver, utils.py:470-502)."""
    dense = [t.numpy() for t in mb.dense()]
    adj, afm, rels = dense[0], dense[1], dense[2:7]
    out = []
    for b in range(mb.B):
        n = int(mb.sizes[b])
        out.append((adj[b, :n, :n].copy(), afm[b, :n].copy()) + tuple(r[b, :, :n, :n].copy() for r in rels) +
                   (mb.labels[b].copy(), 'mol%d' % b, np.zeros((n, 1), dtype=np.float32), b))
    return out


def case(name, fn, mb):
    mols = molecules(mb)
    got = fn(mols)
    data = {'n_mol': np.array(len(mols)), 'channels': np.array(mb.rel_channels)}
    for b, m in enumerate(mols):
        for j, key in enumerate(('adj', 'afm', 'r0', 'r1', 'r2', 'r3', 'r4', 'label')):
            data['in/%d/%s' % (b, key)] = np.asarray(m[j])
    for j, key in enumerate(('adj', 'afm', 'r0', 'r1', 'r2', 'r3', 'r4', 'label', 'subtype', 'size', 'index')):
        data['out/' + key] = got[j].numpy()
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **data)
    print('wrote', name, [tuple(t.shape) for t in got[:3]])


def class_weight_case():
    """utils.py:681-700 ``set_weight`` and utils.py:653-679 ``weight_tensor`` lifted the same way: labels of a synthetic
    training set -> the reference's weight dictionary (stored as [T,2]) and the per-entry weight vector of one batch."""
    tree = ast.parse(open(REF).read())
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ('set_weight', 'weight_tensor')]
    assert len(wanted) == 2
    ns = {'np': np, 'torch': torch, 'use_cuda': False, 'IntTensor': torch.IntTensor}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), REF, 'exec'), ns)
    rng = np.random.default_rng(77)
    T = 6
    y_all = rng.choice([1.0, 0.0, -1.0], size=(400, T), p=[0.08, 0.85, 0.07]).astype(np.float32)
    y_all[:, 4] = rng.choice([1.0, 0.0, -1.0], size=400, p=[0.45, 0.45, 0.10])      # a balanced task
    wd = ns['set_weight'](y_all.tolist())
    assert sorted(wd.keys()) == list(range(T))
    w = np.array([wd[j] for j in range(T)], dtype=np.float64)
    batch = y_all[:37]
    wt = ns['weight_tensor'](wd, torch.from_numpy(batch)).numpy()
    np.savez_compressed(os.path.join(OUT, 'class_weights.npz'), y_all=y_all, weights=w, batch=batch, weight_tensor=wt)
    print('wrote class_weights', w[:2].tolist())


def main():
    class_weight_case()
    cls, reg = reference_collates()
    case('collate_class', cls, make_batch(B=7, n_max=19, n_med=8, rel_channels=(9, 4, 2, 2, 2), seed=31, n_tasks=3,
                                          isolated_frac=0.1, force_max=False))
    case('collate_reg', reg, make_batch(B=5, n_max=14, n_med=6, rel_channels=(9, 4, 2, 2, 2), seed=32, n_tasks=1, task='reg',
                                        force_max=False))


if __name__ == '__main__':
    main()
