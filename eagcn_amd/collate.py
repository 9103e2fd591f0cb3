"""Compact collate: the batch the reference's ``mol_collate_func_class`` / ``mol_collate_func_reg`` (utils.py:504-640) would
deliver, without its dense padding.

The reference pads every molecule's [n,n] adjacency, [C,n,n] relation tensors and [n,24] atom features to the batch maximum on
the host and ships 4 (1 + sum C_k) B N^2 + 96 B N bytes per batch.  ``collate_compact`` takes the SAME per-molecule tuples
(adj, afm, TypeAtt, OrderAtt, AromAtt, ConjAtt, RingAtt, label, smile, subtype, index) and ships O(atoms + bonds): the directed
bond list with the bond type of every attention view (the relation tensors are one-hot at bonded positions, neural_fp.py:111-120),
the unpadded atom-feature rows with per-molecule offsets, sizes and labels.  Padding happens on the device
(``eagcn_pad_rows``), the index comes from ``eagcn_index_from_bonds``; ``EAGCN.forward_compact(bonds, afms, size)`` then gives
the results ``EAGCN.forward`` gives on the reference's dense tensors (tests: golden collate fixture)."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .synthetic import CompactBonds


def compact_host(batch):
    """Host half: list of per-molecule tuples -> dict of numpy arrays (no padding).
    bonds (mol, i, j) int32 [E] in row-major order of each adjacency, codes uint8 [E,K], rows float32 [sum n, F],
    offsets int32 [B+1], sizes int64 [B], labels float32 [B, ...]."""
    bm, bi, bj, codes, rows, sizes, labels = [], [], [], [], [], [], []
    K = 5
    for b, datum in enumerate(batch):
        adj, afm, rels = np.asarray(datum[0]), np.asarray(datum[1], dtype=np.float32), [np.asarray(r) for r in datum[2:7]]
        n = adj.shape[0]
        i, j = np.nonzero(adj)
        bm.append(np.full(i.shape, b, dtype=np.int32))
        bi.append(i.astype(np.int32))
        bj.append(j.astype(np.int32))
        c = np.zeros((len(i), K), dtype=np.uint8)
        for k, r in enumerate(rels):
            hot = r[:, i, j]                                   # [C_k, E_b]
            if len(i) and not ((hot == 1).sum(0) == 1).all():
                raise ValueError('molecule %d, view %d: relation channels are not one-hot at the bonds' % (b, k))
            c[:, k] = hot.argmax(0).astype(np.uint8) if len(i) else 0
        codes.append(c)
        rows.append(afm.reshape(n, -1))
        sizes.append(n)
        labels.append(np.asarray(datum[7], dtype=np.float32))
    off = np.zeros(len(batch) + 1, dtype=np.int32)
    off[1:] = np.cumsum(sizes)
    return {'bond_mol': np.concatenate(bm), 'bond_i': np.concatenate(bi), 'bond_j': np.concatenate(bj),
            'bond_code': np.concatenate(codes), 'rows': np.concatenate(rows), 'offsets': off,
            'sizes': np.asarray(sizes, dtype=np.int64), 'labels': np.stack(labels),
            'channels': [int(np.asarray(r).shape[0]) for r in batch[0][2:7]]}


def pad_rows(rows, offsets, B, N):
    """Device half of the padding: rows [sum n, F] + offsets [B+1] (device tensors) -> [B, N, F]."""
    if not rows.is_cuda:
        raise L.EagcnHipError('pad_rows needs device tensors (no CPU path)')
    F = rows.shape[1]
    out = torch.empty((B, N, F), dtype=torch.float32, device=rows.device)
    L.check(L.load().eagcn_pad_rows(rows.data_ptr(), offsets.data_ptr(), B, N, F, out.data_ptr(),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'eagcn_pad_rows')
    return out


def collate_compact(batch, device, n_pad=None):
    """-> (CompactBonds, afms [B,N,F] padded on the device, size [B], labels): the arguments of EAGCN.forward_compact /
    fused_step(bonds=...).  N = the batch maximum as in the reference (utils.py:583), or ``n_pad`` if given (a fixed N keeps
    one captured graph per model, utils.py:584's commented-out max_molsize)."""
    h = compact_host(batch)
    B = len(batch)
    N = int(n_pad) if n_pad else int(h['sizes'].max())
    if N < int(h['sizes'].max()):
        raise ValueError('n_pad=%d is smaller than the largest molecule (%d atoms)' % (N, int(h['sizes'].max())))

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)
    bonds = CompactBonds(B, N, list(h['channels']), dev(h['bond_mol']), dev(h['bond_i']), dev(h['bond_j']), dev(h['bond_code']))
    afms = pad_rows(dev(h['rows']), dev(h['offsets']), B, N)
    return bonds, afms, dev(h['sizes']), dev(h['labels'])
