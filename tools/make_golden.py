#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference.

Runs only in the build container (needs /root/reference).  The reference's ``layers.py`` imports
as-is; ``models.py`` does ``from utils import *`` which needs RDKit, so an empty stub module named
``utils`` is put in ``sys.modules`` first (EAGCN uses nothing from it).  No reference source or
bytecode is written anywhere; only inputs / parameters / outputs / gradients (data) are saved.

    python tools/make_golden.py            # rewrites tests/golden/*.npz

Each .npz holds:  meta (json), batch/* (compact synthetic batch, see eagcn_amd/synthetic.py),
sd/* (state_dict before the step), x_in (layer cases), gout* (the random cotangents that define
the scalar loss), out/* (forward results), grad/* (parameter / input gradients), sd_after/*
(buffers after one training-mode forward).
"""
import copy
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = '/root/reference/eagcn_pytorch'
sys.path.insert(0, REF)
sys.modules.setdefault('utils', types.ModuleType('utils'))
import layers as ref_layers   # noqa: E402  (reference, read-only)
import models as ref_models   # noqa: E402

from eagcn_amd.synthetic import make_batch   # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
torch.set_num_threads(4)


def init_like_train(model, gen):
    """Deterministic, non-degenerate parameters: reference init ranges, our own generator
    (train.py never seeds torch, so there is no reference RNG stream to match)."""
    for name, p in model.named_parameters():
        if name.endswith('batch_norm.weight') or name.endswith('batch_norm.bias'):
            p.data.zero_()                      # unused, uninitialised in the reference
        elif name.endswith('graph_conv.weight') or name.endswith('graph_conv.W'):
            p.data.normal_(0.0, 0.3, generator=gen)
        elif name.endswith('feature_layer.weight') or name.endswith('adjacent_layer.weight'):   # Diff_Pooling (layers.py:495-496)
            p.data.normal_(0.0, 0.3, generator=gen)
        elif name.endswith('graph_conv.a'):               # GAT attention vector (layers.py:114)
            p.data.normal_(0.0, 0.5, generator=gen)
        elif '.bn.weight' in name or name.startswith('Graph_BN.weight') or 'bn_den' in name and name.endswith('weight'):
            p.data.normal_(1.0, 0.2, generator=gen)
        elif '.bn.bias' in name or name.startswith('Graph_BN.bias') or 'bn_den' in name and name.endswith('bias'):
            p.data.normal_(0.0, 0.2, generator=gen)
        elif name.endswith('self_r'):
            p.data.uniform_(-0.5, 0.5, generator=gen)
        elif name.endswith('att.weight'):
            p.data.uniform_(-1.0, 1.0, generator=gen)
        elif name.endswith('graph_conv.bias'):
            p.data.uniform_(-0.3, 0.3, generator=gen)
        elif name.endswith('ave.weight') or name.endswith('ave_A.weight'):
            p.data.uniform_(-0.6, 0.6, generator=gen)
        elif name.startswith('den'):
            p.data.uniform_(-0.4, 0.4, generator=gen)
        else:
            raise RuntimeError('unhandled parameter ' + name)
    for name, b in model.named_buffers():
        if name.endswith('running_mean'):
            b.normal_(0.0, 0.1, generator=gen)
        elif name.endswith('running_var'):
            b.uniform_(0.5, 1.5, generator=gen)


def pack_batch(mb):
    return {'batch/sizes': mb.sizes, 'batch/edges': mb.edges, 'batch/codes': mb.codes,
            'batch/afm': mb.afm, 'batch/N': np.int64(mb.N),
            'batch/rel_channels': np.array(mb.rel_channels, dtype=np.int64)}


def sd_np(sd, prefix):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def model_case(name, structure, molfp, training, batch_kw, widths1, widths2, dens, nclass,
               n_bfeat, loss='proj', seed=0):
    gen = torch.Generator().manual_seed(1000 + seed)
    mb = make_batch(rel_channels=(n_bfeat, 4, 2, 2, 2), seed=seed, n_tasks=nclass,
                    task='class' if loss == 'bce' else 'reg', **batch_kw)
    adj, afm, r1, r2, r3, r4, r5, size = mb.dense()
    model = ref_models.EAGCN(n_bfeat, 24, *widths1, *widths2, dens[0], dens[1], nclass, 0.0,
                             structure=structure, molfp_mode=molfp)
    init_like_train(model, gen)
    model.train(training)
    data = {}
    data.update(pack_batch(mb))
    data.update(sd_np(model.state_dict(), 'sd/'))
    # per-layer outputs: run the reference layers in sequence on a deep copy, so the running
    # statistics of ``model`` advance exactly once (in the real forward below)
    probe = copy.deepcopy(model)
    probe.train(training)
    x = afm
    with torch.no_grad():
        for li in range(1, 5):
            x, _ = getattr(probe, 'layer%d' % li)(adj, x, r1, r2, r3, r4, r5)
            data['out/layer%d' % li] = x.numpy().copy()
    out, atom_rep, graph_rep = model(adj, afm, r1, r2, r3, r4, r5, size)
    data['out/out'] = out.detach().numpy().copy()
    data['out/atom_rep'] = atom_rep.numpy().copy()
    data['out/graph_rep'] = graph_rep.detach().numpy().copy()
    meta = dict(kind='model', name=name, structure=structure, molfp=molfp, training=training,
                widths1=list(widths1), widths2=list(widths2), dens=list(dens), nclass=nclass,
                n_bfeat=n_bfeat, n_afeat=24, loss=loss, torch=torch.__version__)
    if loss == 'proj':
        g = torch.randn(out.shape, generator=gen)
        g2 = torch.randn(graph_rep.shape, generator=gen) * 0.1
        data['gout'] = g.numpy()
        data['gout_graph_rep'] = g2.numpy()
        scalar = (out * g).sum() + (graph_rep * g2).sum()
    elif loss == 'bce':
        # train.py:326-331 with utils.py:653-679 evaluated by the reference's own arithmetic
        labels = torch.from_numpy(mb.labels)
        bw = [[5000.0 / (300 + 37 * j), 5000.0 / (4100 + 211 * j)] for j in range(nclass)]
        w = torch.zeros(labels.shape)
        for j in range(nclass):
            w[:, j] = (labels[:, j] == 1).float() * bw[j][0] + (labels[:, j] == 0).float() * bw[j][1]
        non_nan = ((labels == 1).sum() + (labels == 0).sum()).float()
        scalar = torch.nn.functional.binary_cross_entropy_with_logits(
            out.view(-1), labels.view(-1), weight=w.view(-1), reduction='sum') / non_nan
        data['labels'] = mb.labels
        data['bce_weight'] = np.array(bw, dtype=np.float32)
    else:
        labels = torch.from_numpy(mb.labels)
        scalar = torch.nn.functional.mse_loss(out.view(-1), labels.view(-1))
        data['labels'] = mb.labels
    data['out/loss'] = scalar.detach().numpy().copy()
    scalar.backward()
    for k, p in model.named_parameters():
        if p.grad is not None:
            data['grad/' + k] = p.grad.numpy().copy()
    data.update(sd_np({k: v for k, v in model.state_dict().items()
                       if 'running' in k or 'num_batches' in k}, 'sd_after/'))
    data['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **data)
    print('wrote', name, 'loss', float(scalar))


def layer_case(name, structure, training, batch_kw, fin, widths, n_bfeat, seed=0, pad_rows_zero=False):
    gen = torch.Generator().manual_seed(2000 + seed)
    mb = make_batch(rel_channels=(n_bfeat, 4, 2, 2, 2), seed=seed, **batch_kw)
    adj, afm, r1, r2, r3, r4, r5, size = mb.dense()
    layer = ref_layers.GraphConv_Layer(fin, n_bfeat, *widths, 0.0, structure)
    init_like_train(layer, gen)
    layer.train(training)
    x_in = torch.rand(mb.B, mb.N, fin, generator=gen) - 0.3
    if pad_rows_zero:
        m = adj.max(dim=2, keepdim=True)[0]
        x_in = x_in * m
    x_in.requires_grad_(True)
    data = {}
    data.update(pack_batch(mb))
    data.update(sd_np(layer.state_dict(), 'sd/'))
    data['x_in'] = x_in.detach().numpy().copy()
    y, a_w = layer(adj, x_in, r1, r2, r3, r4, r5)
    data['out/x'] = y.detach().numpy().copy()
    data['out/A_weight'] = a_w.detach().numpy().copy()
    g = torch.randn(y.shape, generator=gen)
    data['gout'] = g.numpy()
    (y * g).sum().backward()
    data['grad/x_in'] = x_in.grad.numpy().copy()
    for k, p in layer.named_parameters():
        if p.grad is not None:
            data['grad/' + k] = p.grad.numpy().copy()
    data.update(sd_np({k: v for k, v in layer.state_dict().items()
                       if 'running' in k or 'num_batches' in k}, 'sd_after/'))
    meta = dict(kind='layer', name=name, structure=structure, training=training, fin=fin,
                widths=list(widths), n_bfeat=n_bfeat, torch=torch.__version__)
    data['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **data)
    print('wrote', name)


def main():
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]          # optional: names of the cases to (re)generate; default all
    global model_case, layer_case
    if only:
        mc, lc = model_case, layer_case
        model_case = lambda name, *a, **k: mc(name, *a, **k) if name in only else None
        layer_case = lambda name, *a, **k: lc(name, *a, **k) if name in only else None
    small = dict(B=6, n_max=12, n_med=6)
    w1 = (8, 6, 4, 4, 5)
    w2 = (10, 7, 5, 6, 4)
    # --- whole model, reference as shipped (4 layers) ---
    model_case('model_concate_train', 'Concate', 'sum', True, small, w1, w2, (16, 8), 3, 7, seed=1)
    model_case('model_concate_eval', 'Concate', 'sum', False, small, w1, w2, (16, 8), 3, 7, seed=2)
    model_case('model_weighted_train', 'Weighted_sum', 'sum', True, small, (3, 2, 2, 2, 3),
               (4, 3, 2, 2, 3), (16, 8), 2, 5, seed=3)
    model_case('model_weighted_eval', 'Weighted_sum', 'sum', False, small, (3, 2, 2, 2, 3),
               (4, 3, 2, 2, 3), (16, 8), 2, 5, seed=4)
    model_case('model_concate_ave_train', 'Concate', 'ave', True, small, w1, w2, (16, 8), 1, 7, seed=5)
    model_case('model_concate_bce_train', 'Concate', 'sum', True, dict(B=8, n_max=10, n_med=5),
               w1, w2, (16, 8), 4, 7, loss='bce', seed=6)
    model_case('model_concate_mse_train', 'Concate', 'sum', True, dict(B=8, n_max=10, n_med=5),
               w1, w2, (16, 8), 1, 7, loss='mse', seed=7)
    # edge cases of the collate contract
    model_case('model_concate_isolated', 'Concate', 'sum', True,
               dict(B=5, n_max=14, n_med=8, isolated_frac=0.25), w1, w2, (16, 8), 2, 7, seed=8)
    model_case('model_weighted_isolated', 'Weighted_sum', 'sum', True,
               dict(B=5, n_max=14, n_med=8, isolated_frac=0.25), (3, 2, 2, 2, 3), (4, 3, 2, 2, 3),
               (16, 8), 2, 5, seed=9)
    model_case('model_concate_allfull', 'Concate', 'sum', True, dict(B=4, n_max=9, all_full=True),
               w1, w2, (16, 8), 2, 7, seed=10)
    model_case('model_concate_single_atom', 'Concate', 'sum', True,
               dict(B=4, n_max=7, sizes=np.array([1, 7, 3, 1]), force_max=True), w1, w2, (16, 8), 2, 7, seed=11)
    model_case('model_concate_bigN', 'Concate', 'sum', True, dict(B=3, n_max=33, n_med=12),
               (6, 5, 4, 3, 2), (7, 6, 5, 4, 3), (12, 6), 2, 3, seed=12)
    # --- the Kipf-GCN baseline of models.py:63-67 (Vanilla_GCN layers, layers.py:205-258): SURVEY 8 row f-4 ---
    model_case('model_gcn_train', 'GCN', 'sum', True, dict(B=6, n_max=12, n_med=6, isolated_frac=0.15), w1, w2, (16, 8), 3, 7,
               seed=13)
    model_case('model_gcn_eval', 'GCN', 'sum', False, small, w1, w2, (16, 8), 2, 7, seed=14)
    model_case('model_gcn_ave_bce_train', 'GCN', 'ave', True, dict(B=8, n_max=10, n_med=5), w1, w2, (16, 8), 4, 7, loss='bce',
               seed=15)
    # --- the GAT baseline of models.py:69-73 (layers.py:99-203), eval mode only: its attention dropout (0.5, not
    #     configurable) makes training-mode outputs a function of torch's RNG stream ---
    model_case('model_gat_eval', 'GAT', 'sum', False, dict(B=6, n_max=12, n_med=6, isolated_frac=0.15), w1, w2, (16, 8), 3, 7,
               seed=16)
    model_case('model_gat_ave_eval', 'GAT', 'ave', False, small, (4, 3, 3, 2, 2), (5, 4, 3, 2, 2), (16, 8), 2, 7, loss='mse', seed=17)
    # --- molfp_mode='pool' (Diff_Pooling read-out, layers.py:492-506, models.py:90-92, 104-106) over every layer family ---
    iso = dict(B=6, n_max=12, n_med=6, isolated_frac=0.15)
    model_case('model_concate_pool_train', 'Concate', 'pool', True, iso, w1, w2, (16, 8), 3, 7, seed=31)
    model_case('model_weighted_pool_train', 'Weighted_sum', 'pool', True, iso, (3, 2, 2, 2, 3), (4, 3, 2, 2, 3), (16, 8), 2, 5,
               seed=32)
    model_case('model_weighted_pool_eval', 'Weighted_sum', 'pool', False, small, (3, 2, 2, 2, 3), (4, 3, 2, 2, 3), (16, 8), 2, 5,
               seed=33)
    model_case('model_gcn_pool_train', 'GCN', 'pool', True, iso, w1, w2, (16, 8), 3, 7, loss='mse', seed=34)
    model_case('model_gat_pool_eval', 'GAT', 'pool', False, iso, w1, w2, (16, 8), 2, 7, seed=35)
    # --- single layers (used for the 2-/3-layer parity of the n_layers extension) ---
    layer_case('layer_concate_train', 'Concate', True, small, 24, (8, 6, 4, 4, 5), 7, seed=21)
    layer_case('layer_concate_eval', 'Concate', False, small, 24, (8, 6, 4, 4, 5), 7, seed=22)
    layer_case('layer_concate_wide_train', 'Concate', True, dict(B=4, n_max=20, n_med=9), 27,
               (17, 16, 12, 5, 9), 6, seed=23, pad_rows_zero=True)
    layer_case('layer_weighted_train', 'Weighted_sum', True, small, 11, (9, 9, 9, 9, 9), 5, seed=24)
    layer_case('layer_weighted_eval', 'Weighted_sum', False, small, 11, (9, 9, 9, 9, 9), 5, seed=25)
    layer_case('layer_concate_isolated', 'Concate', True,
               dict(B=5, n_max=14, n_med=8, isolated_frac=0.3), 10, (4, 4, 4, 4, 4), 4, seed=26)


if __name__ == '__main__':
    main()
