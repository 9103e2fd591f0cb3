"""Shared test helpers: golden-case loading, oracle construction, comparison metric."""
import glob
import json
import os

import numpy as np
import torch

from eagcn_amd.synthetic import MolBatch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def golden_cases(kind=None):
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, '*.npz'))):
        name = os.path.basename(p)[:-4]
        if kind is None or name.startswith(kind + '_'):
            out.append(name)
    return out


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + '.npz'))
        self.z = z
        self.meta = json.loads(str(z['meta']))
        self.batch = MolBatch(B=len(z['batch/sizes']), N=int(z['batch/N']),
                              n_afeat=z['batch/afm'].shape[2],
                              rel_channels=[int(c) for c in z['batch/rel_channels']],
                              sizes=z['batch/sizes'], edges=z['batch/edges'],
                              codes=z['batch/codes'], afm=z['batch/afm'])

    def group(self, prefix):
        return {k[len(prefix):]: self.z[k] for k in self.z.files if k.startswith(prefix)}

    def state_dict(self):
        return {k: torch.from_numpy(v.copy()) for k, v in self.group('sd/').items()}


def rel_err(a, b):
    """max|a-b| / max|b|  (SURVEY.md 8(d) parity metric); b is the reference."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.numel() == 0:
        return 0.0
    den = b.abs().max().item()
    num = (a - b).abs().max().item()
    return num / den if den > 0 else num


def build_oracle_model(meta, n_layers=4):
    from oracle.eagcn_ref import RefEAGCN
    return RefEAGCN(meta['n_bfeat'], meta['n_afeat'], meta['widths1'], meta['widths2'],
                    meta['dens'][0], meta['dens'][1], meta['nclass'], 0.0,
                    structure=meta['structure'], molfp_mode=meta['molfp'], n_layers=n_layers)


def build_oracle_layer(meta, rel_channels):
    from oracle.eagcn_ref import RefGraphConvLayer
    return RefGraphConvLayer(meta['fin'], rel_channels, meta['widths'], 0.0, meta['structure'])


def assert_grad_close(got, ref, scale, name='', rtol=2e-5, floor=1e-6):
    """|got-ref|_max <= rtol*|ref|_max + floor*scale.

    ``scale`` = largest gradient magnitude over all parameters of the case.  The floor covers
    gradients that are analytically zero (a bias in front of a training-mode BatchNorm receives
    sum(dY) == 0): the reference itself holds only summation noise there (~1e-7 * scale)."""
    got = torch.as_tensor(got, dtype=torch.float64).cpu()
    ref = torch.as_tensor(ref, dtype=torch.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs().max().item() if ref.numel() else 0.0
    bound = rtol * ref.abs().max().item() + floor * float(scale)
    assert err <= bound, (name, err, bound)
