"""Pin the CPU oracle (oracle/eagcn_ref.py) against every golden vector produced by the
unmodified reference (tools/make_golden.py).  fp32 CPU vs fp32 CPU, same ATen op sequence:
tolerance 2e-6 relative (max|d|/max|ref|)."""
import numpy as np
import pytest
import torch

from helpers import Golden, assert_grad_close, build_oracle_layer, build_oracle_model, golden_cases, rel_err

TOL = 2e-6


def _scalar_loss(g, out, graph_rep):
    from oracle.eagcn_ref import classification_loss, regression_loss
    kind = g.meta['loss']
    if kind == 'proj':
        return (out * torch.from_numpy(g.z['gout'])).sum() + \
               (graph_rep * torch.from_numpy(g.z['gout_graph_rep'])).sum()
    labels = torch.from_numpy(g.z['labels'])
    if kind == 'bce':
        return classification_loss(out, labels, g.z['bce_weight'].tolist())
    return regression_loss(out, labels)


@pytest.mark.parametrize('name', golden_cases('model'))
def test_model_matches_reference(name):
    g = Golden(name)
    model = build_oracle_model(g.meta)
    model.load_state_dict(g.state_dict(), strict=True)        # identical key set to the reference
    model.train(g.meta['training'])
    dense = g.batch.dense()
    adj, afm, rels, size = dense[0], dense[1], dense[2:-1], dense[-1]
    import copy
    probe = copy.deepcopy(model)
    with torch.no_grad():
        for i, x in enumerate(probe.layer_outputs(adj, afm, *rels)):
            assert rel_err(x, g.z['out/layer%d' % (i + 1)]) < TOL, 'layer%d' % (i + 1)
    out, atom_rep, graph_rep = model(adj, afm, *rels, size)
    assert rel_err(out, g.z['out/out']) < TOL
    assert rel_err(atom_rep, g.z['out/atom_rep']) < TOL
    assert rel_err(graph_rep, g.z['out/graph_rep']) < TOL
    loss = _scalar_loss(g, out, graph_rep)
    assert abs(float(loss.detach()) - float(g.z['out/loss'])) <= 5e-6 * max(1.0, abs(float(g.z['out/loss'])))
    loss.backward()
    grads = g.group('grad/')
    got = {k: p.grad for k, p in model.named_parameters() if p.grad is not None}
    assert set(got) == set(grads), 'set of parameters that receive a gradient differs'
    scale = max(np.abs(v).max() for v in grads.values())
    for k, ref in grads.items():
        # per-tensor relative error, with an absolute floor for gradients that are analytically
        # zero (biases in front of a training-mode BatchNorm)
        assert_grad_close(got[k], ref, scale, k)
    for k, ref in g.group('sd_after/').items():
        assert rel_err(model.state_dict()[k].double(), ref) < TOL, k


@pytest.mark.parametrize('name', golden_cases('layer'))
def test_layer_matches_reference(name):
    g = Golden(name)
    layer = build_oracle_layer(g.meta, g.batch.rel_channels)
    layer.load_state_dict(g.state_dict(), strict=True)
    layer.train(g.meta['training'])
    dense = g.batch.dense()
    adj, rels = dense[0], dense[2:-1]
    x_in = torch.from_numpy(g.z['x_in'].copy()).requires_grad_(True)
    y, a_w = layer(adj, x_in, *rels)
    assert rel_err(y, g.z['out/x']) < TOL
    assert rel_err(a_w, g.z['out/A_weight']) < TOL
    (y * torch.from_numpy(g.z['gout'])).sum().backward()
    grads = g.group('grad/')
    scale = max(np.abs(v).max() for v in grads.values())
    assert rel_err(x_in.grad, grads.pop('x_in')) < 2e-5
    for k, ref in grads.items():
        p = dict(layer.named_parameters())[k]
        assert_grad_close(p.grad, ref, scale, k)
    for k, ref in g.group('sd_after/').items():
        assert rel_err(layer.state_dict()[k].double(), ref) < TOL, k


def test_fewer_layers_equal_prefix_of_reference_stack():
    """n_layers extension: the first two layers of a 2-layer oracle are the reference's layer1/2."""
    g = Golden('model_concate_train')
    model = build_oracle_model(g.meta, n_layers=2)
    sd = {k: v for k, v in g.state_dict().items()
          if k.startswith('layer1.') or k.startswith('layer2.')}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected
    assert all(not m.startswith('layer') for m in missing)
    dense = g.batch.dense()
    with torch.no_grad():
        outs = model.layer_outputs(dense[0], dense[1], *dense[2:-1])
    assert rel_err(outs[0], g.z['out/layer1']) < TOL
    assert rel_err(outs[1], g.z['out/layer2']) < TOL
