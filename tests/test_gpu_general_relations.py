"""General (non one-hot) relation tensors (reference layers.py:82: the attention logit is a plain 1x1
convolution over the relation channels, S = sum_c w[c] R[c,i,j], for ANY channel values).  The HIP path serves them through
the compact collate: the distinct channel vectors at the bonds become the code book of each view (collate_compact(...,
general=True)), the kernels look sigma(<w, vector>) up by code and turn the per-code gradient histogram back into the
per-channel gradient.  Checked against the CPU oracle on the dense tensors."""
import numpy as np
import pytest
import torch

from eagcn_amd import EAGCN
from eagcn_amd.collate import collate_compact
from eagcn_amd.synthetic import make_batch
from helpers import assert_grad_close, rel_err

pytestmark = pytest.mark.gpu


def _general_molecules(seed, B, channels):
    """Per-molecule tuples whose relation tensors hold, at every bond, one of a handful of random REAL-valued channel vectors
    (symmetric per bond; multi-hot and fractional entries), and the dense padded tensors the reference's collate builds."""
    rng = np.random.default_rng(seed)
    mb = make_batch(B=B, n_max=21, n_med=9, rel_channels=channels, seed=seed, n_tasks=2, task='reg', isolated_frac=0.1)
    dense = [t.numpy() for t in mb.dense()]
    adj, afm = dense[0], dense[1]
    palettes = [np.round(rng.normal(0.0, 1.0, size=(5 + k, c)), 2).astype(np.float32) * (rng.random((5 + k, c)) < 0.6)
                for k, c in enumerate(channels)]
    rels = [np.zeros((B, c, mb.N, mb.N), dtype=np.float32) for c in channels]
    for b in range(B):
        i, j = np.nonzero(np.triu(adj[b]))
        for k, pal in enumerate(palettes):
            pick = pal[rng.integers(0, len(pal), size=len(i))]           # [E, C_k]
            rels[k][b][:, i, j] = pick.T
            rels[k][b][:, j, i] = pick.T
    mols = []
    for b in range(B):
        n = int(mb.sizes[b])
        mols.append((adj[b, :n, :n].copy(), afm[b, :n].copy()) + tuple(r[b, :, :n, :n].copy() for r in rels) +
                    (mb.labels[b].copy(), 'm%d' % b, None, b))
    return mb, [torch.from_numpy(t) for t in [adj, afm] + rels], mols


@pytest.mark.parametrize('graph', [False, True])
@pytest.mark.parametrize('structure', ['Concate', 'Weighted_sum'])
def test_general_relation_tensors_vs_oracle(structure, graph):
    from oracle.eagcn_ref import RefEAGCN, regression_loss
    channels = (6, 4, 3, 2, 2)
    mb, dense, mols = _general_molecules(17, 9, channels)
    size = torch.from_numpy(mb.sizes)
    labels = torch.from_numpy(mb.labels)
    torch.manual_seed(4)
    w1, w2 = [8, 6, 4, 4, 6], [10, 8, 6, 6, 6]
    ref = RefEAGCN(6, 24, w1, w2, 16, 8, 2, 0.0, structure=structure, n_layers=2, rel_channels=list(channels))
    for k in range(5):                                  # attention weights of O(1): the logits must matter
        for l in (ref.layer1, ref.layer2):
            getattr(l, 'block%d' % (k + 1)).att.weight.data.normal_(0.0, 0.8)
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    ref.train()
    out_r, _, gr_r = ref(*dense, size)
    regression_loss(out_r, labels).backward()

    dev = torch.device('cuda', 0)
    m = EAGCN(6, 24, widths1=w1, widths2=w2, n_den1=16, n_den2=8, nclass=2, dropout=0.0, structure=structure, n_layers=2,
              rel_channels=list(channels), graph=graph)
    m.load_state_dict(sd0, strict=True)
    m = m.to(dev).train()
    if graph:                                           # another general batch first: same runner, other code books
        _, _, mols0 = _general_molecules(18, 9, channels)
        b0, a0, s0, _ = collate_compact(mols0, dev, general=True, n_pad=21)
        m.forward_compact(b0, a0, s0)[0].sum().backward()
        sd = {k: v.to(dev) for k, v in sd0.items()}
        m.load_state_dict(sd, strict=True)             # (undo the running-statistics update of that step)
        for p in m.parameters():
            p.grad = None
    bonds, afms, size_c, labels_c = collate_compact(mols, dev, general=True, n_pad=21 if graph else None)
    assert bonds.rel_vectors is not None and all(5 <= v.shape[0] <= 10 for v in bonds.rel_vectors)
    out, _, gr = m.forward_compact(bonds, afms, size_c)
    if graph:
        assert len(m._runners) == 1
    torch.nn.functional.mse_loss(out.view(-1), labels_c.view(-1)).backward()
    assert rel_err(out.detach().cpu(), out_r.detach(), 'out (%s)' % structure) < 1e-5
    assert rel_err(gr.detach().cpu(), gr_r.detach(), 'graph_rep') < 1e-5
    got = dict(m.named_parameters())
    scale = max(p.grad.abs().max().item() for p in ref.parameters() if p.grad is not None)
    for k, p in ref.named_parameters():
        if p.grad is not None:
            assert_grad_close(got[k].grad.cpu(), p.grad.numpy(), scale, k, rtol=2e-5, floor=2e-6)
    # the same batch through the reference's DENSE signature: relations='general' builds the code books from the padded tensors
    mg = EAGCN(6, 24, widths1=w1, widths2=w2, n_den1=16, n_den2=8, nclass=2, dropout=0.0, structure=structure, n_layers=2,
               rel_channels=list(channels), graph=graph, relations='general')
    mg.load_state_dict(sd0, strict=True)
    mg = mg.to(dev).train()
    out2, _, gr2 = mg(*[t.to(dev) for t in dense], size.to(dev))
    torch.nn.functional.mse_loss(out2.view(-1), labels_c.view(-1)).backward()
    assert rel_err(out2.detach().cpu(), out_r.detach(), 'out, dense signature (%s)' % structure) < 1e-5
    assert rel_err(gr2.detach().cpu(), gr_r.detach(), 'graph_rep, dense signature') < 1e-5
    got2 = dict(mg.named_parameters())
    for k, p in ref.named_parameters():
        if p.grad is not None:
            assert_grad_close(got2[k].grad.cpu(), p.grad.numpy(), scale, k + ' (dense signature)', rtol=2e-5, floor=2e-6)
    # non one-hot batches through the dense signature of a default model are refused loudly, with the way out
    with pytest.raises(Exception, match='one-hot'):
        m(*[t.to(dev) for t in dense], size.to(dev))
    with pytest.raises(ValueError, match='general=True'):
        collate_compact(mols, dev)
