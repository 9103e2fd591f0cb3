"""Device-side collate: EAGCN.forward on the tensors the REFERENCE's collate built (fixture) and
EAGCN.forward_compact / fused_step on eagcn_amd.collate.collate_compact of the same molecules give identical results, and
the on-device padding equals the reference's."""
import copy

import numpy as np
import pytest
import torch

from eagcn_amd import EAGCN
from eagcn_amd.collate import collate_compact
from eagcn_amd.losses import fused_classification_loss
from eagcn_amd.synthetic import bce_weights
from test_collate_cpu import _load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('graph', [False, True])
def test_compact_collate_equals_reference_collate_through_the_model(graph):
    dev = torch.device('cuda', 0)
    z, mols = _load('collate_class')
    dense = [torch.from_numpy(z['out/' + k]).to(dev) for k in ('adj', 'afm', 'r0', 'r1', 'r2', 'r3', 'r4')]
    size = torch.from_numpy(z['out/size']).to(dev)
    labels_ref = torch.from_numpy(z['out/label']).to(dev)
    bonds, afms, size_c, labels = collate_compact(mols, dev)
    assert torch.equal(afms, dense[1])                                  # eagcn_pad_rows == utils.py:588-590
    assert torch.equal(size_c, size) and torch.equal(labels.view_as(labels_ref), labels_ref)
    torch.manual_seed(3)
    a = EAGCN(9, 24, widths1=[8] * 5, widths2=[12] * 5, n_den1=16, n_den2=8, nclass=3, dropout=0.0, n_layers=2,
              rel_channels=[9, 4, 2, 2, 2], graph=graph).to(dev).train()
    b = copy.deepcopy(a)
    bw = torch.tensor(bce_weights(3), dtype=torch.float32, device=dev)
    out_a, _, gr_a = a(*dense, size)
    fused_classification_loss(out_a, labels_ref, bw).backward()
    out_b, _, gr_b = b.forward_compact(bonds, afms, size_c)
    fused_classification_loss(out_b, labels, bw).backward()
    assert torch.equal(out_a, out_b) and torch.equal(gr_a, gr_b)
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        if p.grad is not None:
            assert torch.equal(p.grad, q.grad), n
    if graph:                                  # and the whole step as one graph from the compact batch
        c = copy.deepcopy(a)
        for p in c.parameters():
            p.grad = None
        # (a's running statistics advanced once above; c starts from them: compare with a second step of a)
        for p in a.parameters():
            p.grad = None
        out_a2, _, _ = a(*dense, size)
        la = fused_classification_loss(out_a2, labels_ref, bw)
        la.backward()
        lc, (out_c, _, _) = c.fused_step((afms, size_c), labels, 'class', bw, bonds=bonds)
        assert torch.equal(out_a2, out_c) and torch.equal(la.detach(), lc)
        for (n, p), q in zip(a.named_parameters(), c.parameters()):
            if p.grad is not None:
                assert torch.equal(p.grad, q.grad), n


def test_fixed_padding_width():
    dev = torch.device('cuda', 0)
    z, mols = _load('collate_reg')
    bonds, afms, size, _ = collate_compact(mols, dev, n_pad=16)
    assert afms.shape[1] == 16 and bonds.N == 16
    ref = np.zeros((len(mols), 16, 24), dtype=np.float32)
    ref[:, :z['out/afm'].shape[1]] = z['out/afm']
    assert np.array_equal(afms.cpu().numpy(), ref)
    with pytest.raises(ValueError):
        collate_compact(mols, dev, n_pad=3)
