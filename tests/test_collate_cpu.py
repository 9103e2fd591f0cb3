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
