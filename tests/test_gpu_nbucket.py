"""EAGCN(graph=True, n_bucket=16): batches padded to different N (the reference pads every batch to ITS maximum,
utils.py:583) share ONE captured runner per bucket, and give the results of the eager engine on the batch's own N --
BatchNorm row counts B*N, the 1e-9 filler weights and the padding-row terms follow the batch, not the capacity."""
import copy

import pytest
import torch

from eagcn_amd import EAGCN
from eagcn_amd.losses import fused_regression_loss
from eagcn_amd.synthetic import make_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('structure', ['Concate', 'Weighted_sum'])
def test_bucketed_runner_equals_eager_on_the_batch_own_padding(structure):
    dev = torch.device('cuda', 0)
    torch.manual_seed(9)
    kw = dict(widths1=[8] * 5, widths2=[12] * 5, n_den1=16, n_den2=8, nclass=1, dropout=0.25, structure=structure, n_layers=2)
    g = EAGCN(28, 24, graph=True, n_bucket=16, **kw).to(dev).train()
    e = EAGCN(28, 24, graph=False, **kw)
    e.load_state_dict(copy.deepcopy(g.state_dict()))
    e = e.to(dev).train()
    for step, n_max in enumerate((19, 30, 23, 17, 32, 19)):          # all in the bucket (16, 32]
        mb = make_batch(B=10, n_max=n_max, n_med=8, rel_channels=(28, 4, 2, 2, 2), seed=70 + step, n_tasks=1, task='reg',
                        isolated_frac=0.1)
        assert mb.N == n_max
        dense = tuple(t.to(dev) for t in mb.dense())
        labels = torch.from_numpy(mb.labels).to(dev)
        outs = []
        for m in (g, e):
            for p in m.parameters():
                p.grad = None
            torch.manual_seed(1000 + step)                           # the same dropout seeds for both models
            out, atom_rep, gr = m(*dense)
            fused_regression_loss(out, labels).backward()
            outs.append((out.detach().clone(), gr.detach().clone(), atom_rep.cpu().clone(),
                         {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
        (og, gg, ag, pg), (oe, ge, ae, pe) = outs
        assert ag.shape == ae.shape == (10, n_max, ae.shape[2])
        for a, b, name in ((og, oe, 'out'), (gg, ge, 'graph_rep'), (ag, ae, 'atom_rep')):
            d = (a - b).abs().max().item()
            assert d <= 2e-6 * max(b.abs().max().item(), 1e-6), (step, name, d)
        scale = max(v.abs().max().item() for v in pe.values())
        for n in pe:
            d = (pg[n] - pe[n]).abs().max().item()
            assert d <= 2e-6 * scale + 2e-6 * pe[n].abs().max().item(), (step, n, d)
    assert len(g._runners) == 1, list(g._runners)                    # one bucket, one pair of graphs
    sg, se = g.state_dict(), e.state_dict()
    for k in se:
        if 'running' in k:
            assert (sg[k] - se[k]).abs().max().item() <= 2e-6 * max(se[k].abs().max().item(), 1.0), k
