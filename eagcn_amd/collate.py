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


def compact_host(batch, general=False):
    """Host half: list of per-molecule tuples -> dict of numpy arrays (no padding).
    bonds (mol, i, j) int32 [E] in row-major order of each adjacency, codes uint8 [E,K], rows float32 [sum n, F],
    offsets int32 [B+1], sizes int64 [B], labels float32 [B, ...].
    general=False: the relation tensors must be one-hot at the bonds (what neural_fp.py:111-120 produces); code = channel.
    general=True: ANY channel values (layers.py:82 is a plain 1x1 convolution): the distinct channel vectors found at the bonds
    of the batch become the code book of each view (`rel_vectors[k]`, lexicographically sorted, at most 255 per view)."""
    bm, bi, bj, vecs, rows, sizes, labels = [], [], [], [], [], [], []
    K = 5
    for b, datum in enumerate(batch):
        adj, afm, rels = np.asarray(datum[0]), np.asarray(datum[1], dtype=np.float32), [np.asarray(r) for r in datum[2:7]]
        n = adj.shape[0]
        i, j = np.nonzero(adj)
        bm.append(np.full(i.shape, b, dtype=np.int32))
        bi.append(i.astype(np.int32))
        bj.append(j.astype(np.int32))
        vecs.append([np.ascontiguousarray(r[:, i, j].T, dtype=np.float32) for r in rels])     # per view [E_b, C_k]
        rows.append(afm.reshape(n, -1))
        sizes.append(n)
        labels.append(np.asarray(datum[7], dtype=np.float32))
    E = int(sum(len(x) for x in bm))
    codes = np.zeros((E, K), dtype=np.uint8)
    channels, rel_vectors = [], []
    for k in range(K):
        v = np.concatenate([m[k] for m in vecs]) if E else np.zeros((0, np.asarray(batch[0][2 + k]).shape[0]), np.float32)
        if not general:
            if E and not (((v == 1).sum(1) == 1) & ((v != 0).sum(1) == 1)).all():
                raise ValueError('view %d: relation channels are not one-hot at the bonds (use general=True)' % k)
            codes[:, k] = v.argmax(1).astype(np.uint8) if E else 0
            channels.append(int(v.shape[1]))
        else:
            table, inv = np.unique(v, axis=0, return_inverse=True) if E else (np.zeros((1, v.shape[1]), np.float32), np.zeros(0, int))
            if len(table) > 255:
                raise ValueError('view %d: %d distinct relation vectors at the bonds of this batch (at most 255)' % (k, len(table)))
            codes[:, k] = np.asarray(inv).reshape(-1).astype(np.uint8)
            channels.append(int(len(table)))
            rel_vectors.append(np.ascontiguousarray(table, dtype=np.float32))
    off = np.zeros(len(batch) + 1, dtype=np.int32)
    off[1:] = np.cumsum(sizes)
    return {'bond_mol': np.concatenate(bm), 'bond_i': np.concatenate(bi), 'bond_j': np.concatenate(bj),
            'bond_code': codes, 'rows': np.concatenate(rows), 'offsets': off,
            'sizes': np.asarray(sizes, dtype=np.int64), 'labels': np.stack(labels),
            'channels': channels, 'rel_vectors': rel_vectors if general else None}


def bonds_from_dense(adjs, rels, general=True):
    """The reference's padded DEVICE tensors (adjs [B,N,N] in {0,1}, rels[k] [B,C_k,N,N]) -> CompactBonds, for relation tensors
    with arbitrary channel values (layers.py:82): the distinct channel vectors at the bonds become the code book of each view
    (lexicographically sorted, at most 255), exactly as ``compact_host(general=True)`` builds them from per-molecule tuples.
    Input canonicalisation with tensor ops (nonzero / gather / unique; one host sync for the sizes); the compute stays on the
    kernels behind ``EAGCN.forward_compact``.  general=False: one-hot channels, code = channel index."""
    if not bool(((adjs == 0) | (adjs == 1)).all()):
        raise L.EagcnHipError('adjacency entries must be 0 or 1')
    B, N, _ = adjs.shape
    nz = (adjs != 0).nonzero()                                        # [E,3], row-major: the order of np.nonzero per molecule
    bm, bi, bj = nz[:, 0], nz[:, 1], nz[:, 2]
    E = int(nz.shape[0])
    codes = torch.zeros((E, len(rels)), dtype=torch.uint8, device=adjs.device)
    channels, tables = [], []
    for k, r in enumerate(rels):
        v = r.to(torch.float32)[bm, :, bi, bj]                        # [E, C_k]
        if general:
            if E:
                table, inv = torch.unique(v, dim=0, return_inverse=True)
            else:
                table, inv = torch.zeros((1, r.shape[1]), dtype=torch.float32, device=adjs.device), None
            if table.shape[0] > 255:
                raise ValueError('view %d: %d distinct relation vectors at the bonds of this batch (at most 255)'
                                 % (k, table.shape[0]))
            if E:
                codes[:, k] = inv.reshape(-1).to(torch.uint8)
            channels.append(int(table.shape[0]))
            tables.append(table.contiguous())
        else:
            if E and not bool((((v == 1).sum(1) == 1) & ((v != 0).sum(1) == 1)).all()):
                raise ValueError('view %d: relation channels are not one-hot at the bonds (use general=True)' % k)
            if E:
                codes[:, k] = v.argmax(1).to(torch.uint8)
            channels.append(int(r.shape[1]))
    return CompactBonds(B, N, channels, bm.to(torch.int32).contiguous(), bi.to(torch.int32).contiguous(),
                        bj.to(torch.int32).contiguous(), codes, tables if general else None)


def pad_rows(rows, offsets, B, N):
    """Device half of the padding: rows [sum n, F] + offsets [B+1] (device tensors) -> [B, N, F]."""
    if not rows.is_cuda:
        raise L.EagcnHipError('pad_rows needs device tensors (no CPU path)')
    F = rows.shape[1]
    out = torch.empty((B, N, F), dtype=torch.float32, device=rows.device)
    L.check(L.load().eagcn_pad_rows(rows.data_ptr(), offsets.data_ptr(), B, N, F, out.data_ptr(),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'eagcn_pad_rows')
    return out


def collate_compact(batch, device, n_pad=None, general=False):
    """-> (CompactBonds, afms [B,N,F] padded on the device, size [B], labels): the arguments of EAGCN.forward_compact /
    fused_step(bonds=...).  N = the batch maximum as in the reference (utils.py:583), or ``n_pad`` if given (a fixed N keeps
    one captured graph per model, utils.py:584's commented-out max_molsize)."""
    h = compact_host(batch, general)
    B = len(batch)
    N = int(n_pad) if n_pad else int(h['sizes'].max())
    if N < int(h['sizes'].max()):
        raise ValueError('n_pad=%d is smaller than the largest molecule (%d atoms)' % (N, int(h['sizes'].max())))

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)
    bonds = CompactBonds(B, N, list(h['channels']), dev(h['bond_mol']), dev(h['bond_i']), dev(h['bond_j']), dev(h['bond_code']),
                         None if h['rel_vectors'] is None else [dev(v) for v in h['rel_vectors']])
    afms = pad_rows(dev(h['rows']), dev(h['offsets']), B, N)
    return bonds, afms, dev(h['sizes']), dev(h['labels'])
