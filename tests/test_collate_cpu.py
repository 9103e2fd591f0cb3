"""Compact collate (eagcn_amd/collate.py) against the reference's own collate functions (fixtures of
tools/make_collate_golden.py: utils.py:504-640 lifted out with ast and run on synthetic per-molecule data): the compact
arrays, expanded densely, are exactly the tensors the reference pads."""
import os

import numpy as np
import pytest

from eagcn_amd.collate import compact_host
from helpers import GOLDEN


def _load(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    mols = []
    for b in range(int(z['n_mol'])):
        g = lambda k: z['in/%d/%s' % (b, k)]
        mols.append((g('adj'), g('afm'), g('r0'), g('r1'), g('r2'), g('r3'), g('r4'), g('label'), 'mol%d' % b, None, b))
    return z, mols


@pytest.mark.parametrize('name', ['collate_class', 'collate_reg'])
def test_compact_collate_expands_to_the_reference_collate(name):
    z, mols = _load(name)
    h = compact_host(mols)
    B, N = z['out/adj'].shape[0], z['out/adj'].shape[1]
    assert h['sizes'].tolist() == z['out/size'].tolist()
    assert N == int(h['sizes'].max())                                  # utils.py:583 / 529: pad to the batch maximum
    adj = np.zeros((B, N, N), dtype=np.float32)
    adj[h['bond_mol'], h['bond_i'], h['bond_j']] = 1.0
    assert np.array_equal(adj, z['out/adj'])
    for k, c in enumerate(h['channels']):
        r = np.zeros((B, c, N, N), dtype=np.float32)
        r[h['bond_mol'], h['bond_code'][:, k], h['bond_i'], h['bond_j']] = 1.0
        assert np.array_equal(r, z['out/r%d' % k]), 'relation tensor %d' % k
    afm = np.zeros((B, N, h['rows'].shape[1]), dtype=np.float32)
    for b in range(B):
        afm[b, :h['sizes'][b]] = h['rows'][h['offsets'][b]:h['offsets'][b + 1]]
    assert np.array_equal(afm, z['out/afm'])
    assert np.array_equal(h['labels'].reshape(z['out/label'].shape), z['out/label'])


def test_compact_collate_rejects_non_one_hot_relations():
    z, mols = _load('collate_class')
    bad = list(mols[0])
    r = bad[2].copy()
    i, j = np.nonzero(bad[0])
    r[:, i[0], j[0]] = 0.5
    bad[2] = r
    with pytest.raises(ValueError, match='one-hot'):
        compact_host([tuple(bad)] + mols[1:])


def test_general_collate_code_book_reconstructs_the_relation_tensors():
    """general=True: any channel values at the bonds (layers.py:82 is a plain 1x1 convolution); the per-view code book of the
    batch + the codes reproduce the dense relation tensors exactly."""
    z, mols = _load('collate_class')
    rng = np.random.default_rng(3)
    gen = []
    for m in mols:
        adj = m[0]
        i, j = np.nonzero(np.triu(adj))
        rels = []
        for r in m[2:7]:
            pal = np.round(rng.normal(size=(4, r.shape[0])), 1).astype(np.float32)
            g = np.zeros_like(r)
            pick = pal[rng.integers(0, 4, size=len(i))]
            g[:, i, j] = pick.T
            g[:, j, i] = pick.T
            rels.append(g)
        gen.append((adj, m[1]) + tuple(rels) + m[7:])
    h = compact_host(gen, general=True)
    assert h['rel_vectors'] is not None and all(1 <= len(t) <= 255 for t in h['rel_vectors'])
    assert h['channels'] == [len(t) for t in h['rel_vectors']]
    B, N = len(gen), int(h['sizes'].max())
    for k, table in enumerate(h['rel_vectors']):
        assert np.array_equal(table, np.unique(table, axis=0))              # sorted, distinct: a deterministic code book
        dense = np.zeros((B, table.shape[1], N, N), dtype=np.float32)
        dense[h['bond_mol'], :, h['bond_i'], h['bond_j']] = table[h['bond_code'][:, k]]
        for b, m in enumerate(gen):
            n = m[0].shape[0]
            assert np.array_equal(dense[b, :, :n, :n], m[2 + k]), (b, k)
    with pytest.raises(ValueError, match='general=True'):
        compact_host(gen)


def test_bonds_from_dense_equals_compact_host_for_general_relations():
    """The dense-signature canonicalisation (EAGCN(relations='general').forward) builds the same bond list, codes and code
    books from the reference's padded tensors as compact_host(general=True) builds from the per-molecule tuples."""
    import torch
    from eagcn_amd.collate import bonds_from_dense, compact_host
    from test_gpu_general_relations import _general_molecules
    channels = (6, 4, 3, 2, 2)
    mb, dense, mols = _general_molecules(17, 9, channels)
    h = compact_host(mols, general=True)
    b = bonds_from_dense(dense[0], dense[2:], general=True)
    assert (b.bond_mol.numpy() == h['bond_mol']).all() and (b.bond_i.numpy() == h['bond_i']).all()
    assert (b.bond_j.numpy() == h['bond_j']).all()
    assert b.channels == h['channels']
    assert (b.bond_code.numpy() == h['bond_code']).all()
    for t, ref in zip(b.rel_vectors, h['rel_vectors']):
        assert (t.numpy() == ref).all()
    onehot = mb.dense()
    c = bonds_from_dense(onehot[0], onehot[2:-1], general=False)
    assert c.rel_vectors is None and c.channels == list(channels)
    with pytest.raises(ValueError, match='general=True'):
        bonds_from_dense(dense[0], dense[2:], general=False)
