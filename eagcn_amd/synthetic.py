"""Synthetic molecule batches obeying the reference's collate contract.

Contract reproduced (reference: eagcn_pytorch/utils.py:575-640 ``mol_collate_func_class`` and
neural_fp.py:57-122 ``dump_as_matrices_Att``): every molecule is padded to the batch maximum N;
``adj [B,N,N]`` is a symmetric 0/1 float32 matrix with zero diagonal; ``afm [B,N,n_afeat]`` holds
features in [0,1) with zero pad rows; each relation tensor ``[B,C_k,N,N]`` is one-hot over the
channel axis at bonded (i,j) and all-zero elsewhere; ``size [B]`` is int64.

Graph shape follows SURVEY.md section 8(d): atom counts ~ clipped log-normal around the dataset
median with one molecule forced to N_max, a random spanning tree plus floor(n/10) ring closures.
RDKit is not available, so real SMILES cannot be featurised; these are shape-faithful stand-ins.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

# (median atoms, N_max) per dataset family, SURVEY.md 8(d)
DATASET_SHAPES = {'tox21': (16, 132), 'hiv': (23, 222), 'lipo': (27, 115), 'freesolv': (8, 24)}


@dataclass
class MolBatch:
    """Compact host description of a batch + dense materialisation."""
    B: int
    N: int
    n_afeat: int
    rel_channels: List[int]
    sizes: np.ndarray                     # [B] int64 true atom counts
    edges: np.ndarray                     # [E,3] int64 (b,i,j) directed, both directions present
    codes: np.ndarray                     # [E,K] int64 bond type per view
    afm: np.ndarray                       # [B,N,n_afeat] float32
    labels: Optional[np.ndarray] = None   # [B,T] float32 (classification: 1/0/-1) or regression
    _dense: dict = field(default_factory=dict, repr=False)

    def dense(self, device='cpu', dtype=torch.float32):
        """(adj, afm, rel_1..rel_K, size) as the reference's collate would deliver them."""
        key = (str(device), dtype)
        if key in self._dense:
            return self._dense[key]
        B, N = self.B, self.N
        e = torch.from_numpy(self.edges).to(device)
        adj = torch.zeros(B, N, N, dtype=dtype, device=device)
        if e.numel():
            adj[e[:, 0], e[:, 1], e[:, 2]] = 1.0
        rels = []
        cod = torch.from_numpy(self.codes).to(device)
        for k, c in enumerate(self.rel_channels):
            r = torch.zeros(B, c, N, N, dtype=dtype, device=device)
            if e.numel():
                r[e[:, 0], cod[:, k], e[:, 1], e[:, 2]] = 1.0
            rels.append(r)
        afm = torch.from_numpy(self.afm).to(device=device, dtype=dtype)
        size = torch.from_numpy(self.sizes).to(device)
        out = (adj, afm, *rels, size)
        self._dense[key] = out
        return out

    def drop_cache(self):
        self._dense.clear()

    def compact(self, device):
        """(CompactBonds, afm, size): the same molecules without the dense adjacency / relation tensors."""
        e = self.edges
        def dev(a, dt):
            return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)
        bonds = CompactBonds(self.B, self.N, list(self.rel_channels), dev(e[:, 0], torch.int32), dev(e[:, 1], torch.int32),
                             dev(e[:, 2], torch.int32), dev(self.codes, torch.uint8))
        afm = torch.from_numpy(self.afm).to(device=device, dtype=torch.float32)
        return bonds, afm, torch.from_numpy(self.sizes).to(device)


@dataclass
class CompactBonds:
    """Compact description of a batch's bonds for ``EAGCN.forward_compact`` / ``BatchIndex.from_bonds``:
    every DIRECTED bond exactly once (so both (i,j) and (j,i) are listed), int32 molecule / row / column
    indices and the bond's type per attention view as uint8 [E,K] (values < channels[k])."""
    B: int
    N: int
    channels: List[int]
    bond_mol: torch.Tensor
    bond_i: torch.Tensor
    bond_j: torch.Tensor
    bond_code: torch.Tensor
    # general (non one-hot) relation tensors, layers.py:82: code c of view k stands for the channel vector rel_vectors[k][c]
    # (float32 device tensor [channels[k], C_k]); None = one-hot, code c stands for channel c
    rel_vectors: Optional[List[torch.Tensor]] = None

    def first_view(self):
        """The same bonds with the first attention view only (structure='GCN' needs the bond positions, no types)."""
        if len(self.channels) == 1:
            return self
        return CompactBonds(self.B, self.N, list(self.channels[:1]), self.bond_mol, self.bond_i, self.bond_j,
                            self.bond_code[:, :1].contiguous(), None if self.rel_vectors is None else self.rel_vectors[:1])

    def checked(self):
        E = self.bond_mol.numel()
        for t, dt, name in ((self.bond_mol, torch.int32, 'bond_mol'), (self.bond_i, torch.int32, 'bond_i'),
                            (self.bond_j, torch.int32, 'bond_j'), (self.bond_code, torch.uint8, 'bond_code')):
            if t.dtype != dt or not t.is_contiguous() or not t.is_cuda:
                raise ValueError('%s must be a contiguous %s device tensor' % (name, dt))
        if self.bond_i.numel() != E or self.bond_j.numel() != E or tuple(self.bond_code.shape) != (E, len(self.channels)):
            raise ValueError('bond arrays disagree in length')
        return self.bond_mol, self.bond_i, self.bond_j, self.bond_code


def _sample_sizes(rng, B, n_med, n_max, sigma=0.45, force_max=True, n_min=2):
    n = np.rint(rng.lognormal(np.log(n_med), sigma, size=B)).astype(np.int64)
    n = np.clip(n, n_min, n_max)
    if force_max and B > 0:
        n[rng.integers(0, B)] = n_max
    return n


def make_batch(B, n_max, n_med=None, n_afeat=24, rel_channels=(28, 4, 2, 2, 2), seed=1234,
               sizes=None, all_full=False, isolated_frac=0.0, n_tasks=0, task='class',
               force_max=True) -> MolBatch:
    """Build one synthetic batch.

    isolated_frac: fraction of atoms (inside molecules) whose bonds are all removed, to exercise
    the reference's "row mask is 0 for isolated atoms as well as padding" behaviour (layers.py:295).
    """
    rng = np.random.default_rng(seed)
    K = len(rel_channels)
    if sizes is None:
        if all_full:
            sizes = np.full(B, n_max, dtype=np.int64)
        else:
            sizes = _sample_sizes(rng, B, n_med if n_med else max(2, n_max // 3), n_max,
                                  force_max=force_max)
    sizes = np.asarray(sizes, dtype=np.int64)
    assert sizes.shape == (B,) and sizes.max(initial=0) <= n_max
    N = int(sizes.max(initial=1)) if not force_max else n_max
    eb, ei, ej = [], [], []
    for b in range(B):
        n = int(sizes[b])
        if n < 2:
            continue
        # random recursive tree: node v attaches to a uniformly random earlier node
        par = (rng.random(n - 1) * np.arange(1, n)).astype(np.int64)
        src = np.arange(1, n, dtype=np.int64)
        pairs = {(int(min(a, c)), int(max(a, c))) for a, c in zip(src, par)}
        for _ in range(n // 10):
            a, c = rng.integers(0, n, size=2)
            if a != c:
                pairs.add((int(min(a, c)), int(max(a, c))))
        pr = np.array(sorted(pairs), dtype=np.int64)
        if isolated_frac > 0:
            iso = rng.random(n) < isolated_frac
            keep = ~(iso[pr[:, 0]] | iso[pr[:, 1]])
            pr = pr[keep]
        if pr.size == 0:
            continue
        eb.append(np.full(2 * len(pr), b, dtype=np.int64))
        ei.append(np.concatenate([pr[:, 0], pr[:, 1]]))
        ej.append(np.concatenate([pr[:, 1], pr[:, 0]]))
    if eb:
        edges = np.stack([np.concatenate(eb), np.concatenate(ei), np.concatenate(ej)], axis=1)
    else:
        edges = np.zeros((0, 3), dtype=np.int64)
    E = len(edges)
    # one random type per undirected edge and view, symmetric
    codes = np.zeros((E, K), dtype=np.int64)
    if E:
        lo = np.minimum(edges[:, 1], edges[:, 2])
        hi = np.maximum(edges[:, 1], edges[:, 2])
        key = (edges[:, 0] * N + lo) * N + hi
        uniq, inv = np.unique(key, return_inverse=True)
        for k, c in enumerate(rel_channels):
            codes[:, k] = rng.integers(0, c, size=len(uniq))[inv]
    afm = np.zeros((B, N, n_afeat), dtype=np.float32)
    for b in range(B):
        n = int(sizes[b])
        afm[b, :n, :] = rng.random((n, n_afeat), dtype=np.float32)
    labels = None
    if n_tasks:
        if task == 'class':
            labels = rng.choice(np.array([0.0, 1.0, -1.0], dtype=np.float32), size=(B, n_tasks),
                                p=[0.85, 0.08, 0.07]).astype(np.float32)
        else:
            labels = rng.standard_normal((B, n_tasks)).astype(np.float32)
    return MolBatch(B=B, N=N, n_afeat=n_afeat, rel_channels=list(rel_channels), sizes=sizes,
                    edges=edges, codes=codes, afm=afm, labels=labels)


def bce_weights(n_tasks, seed=7):
    """Stand-in for utils.py:681-700 ``set_weight``: [pos_weight, neg_weight] per task."""
    rng = np.random.default_rng(seed)
    pos = rng.integers(200, 900, size=n_tasks)
    neg = rng.integers(4000, 6500, size=n_tasks)
    return [[5000.0 / p, 5000.0 / n] for p, n in zip(pos, neg)]
