#!/usr/bin/env python3
"""The reference training loop (train.py:289-346) on the HIP engine with synthetic molecules: Adam with weight decay,
HIP-graph replay of the step, periodic evaluation (per-task ROC-AUC / RMSE on device-resident buffers).

    python examples/train_synth.py --dataset tox21 --steps 200 --batch 256
    python examples/train_synth.py --dataset tox21 --loader compact      # per-molecule data -> compact collate, every batch
                                                                        # padded to ITS maximum (utils.py:583), n_bucket=16
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eagcn_amd import EAGCN, training, weights_init  # noqa: E402
from eagcn_amd.collate import collate_compact  # noqa: E402
from eagcn_amd.synthetic import make_batch  # noqa: E402

# train.py:61-114: widths, head sizes, learning rate, weight decay, task kind, tasks, (median atoms, N_max)
DATASETS = {'tox21': ([80] * 5, [140] * 5, 256, 64, 5e-4, 1e-4, 'class', 12, 28, (16, 132)),
            'hiv': ([100] * 5, [250] * 5, 512, 128, 1e-3, 1e-5, 'class', 1, 28, (23, 222)),
            'lipo': ([60] * 5, [100] * 5, 128, 64, 1e-4, 1e-3, 'reg', 1, 18, (27, 115)),
            'freesolv': ([40] * 5, [60] * 5, 128, 64, 1e-4, 1e-2, 'reg', 1, 17, (8, 24))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--dataset', default='tox21', choices=sorted(DATASETS))
    ap.add_argument('--arch', default='Concate', choices=('Concate', 'Weighted_sum', 'GCN', 'GAT'),
                    help="models.py:33-73; 'GCN' / 'GAT' are the reference's baselines")
    ap.add_argument('--molfp', default='sum', choices=('sum', 'ave', 'pool'), help='read-out, models.py:104-111')
    ap.add_argument('--layers', type=int, default=2)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--dr', type=float, default=0.3)
    ap.add_argument('--print-freq', type=int, default=50)
    ap.add_argument('--n-train', type=int, default=8, help='distinct synthetic training batches (cycled)')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of HIP-graph replay of the step')
    ap.add_argument('--optimizer', default='flat', choices=('flat', 'torch'),
                    help="train.py:303 Adam(lr, weight_decay): 'flat' = eagcn_amd.optim.FlatAdam (one kernel over a flat parameter "
                         "buffer, inside the captured step; models with a model-level plan), 'torch' = torch.optim.Adam")
    ap.add_argument('--loader', default='dense', choices=('dense', 'compact'),
                    help="dense: the reference's collate tensors; compact: bond list + unpadded rows (collate_compact)")
    args = ap.parse_args()
    w1, w2, d1, d2, lr, wd, task, T, n_bfeat, (n_med, n_max) = DATASETS[args.dataset]
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)

    def batch(seed):
        compact = args.loader == 'compact'
        mb = make_batch(B=args.batch, n_max=n_max, n_med=n_med, rel_channels=(n_bfeat, 4, 2, 2, 2), seed=seed, n_tasks=T,
                        task=task, force_max=not compact)
        if not compact:
            return tuple(mb.dense(dev)), torch.from_numpy(mb.labels).to(dev)
        # what a Dataset.__getitem__ of the reference yields (utils.py:470-502), one tuple per molecule
        dense = [t.numpy() for t in mb.dense()]
        mols = []
        for b in range(mb.B):
            n = int(mb.sizes[b])
            mols.append((dense[0][b, :n, :n], dense[1][b, :n]) + tuple(r[b, :, :n, :n] for r in dense[2:7]) +
                        (mb.labels[b], 'mol%d' % b, None, b))
        bonds, afms, size, labels = collate_compact(mols, dev)
        return (bonds, afms, size), labels
    train_set = [batch(100 + i) for i in range(args.n_train)]
    val_set = [batch(900 + i) for i in range(2)]
    bce_w = None
    if task == 'class':
        bce_w = torch.tensor(training.set_weight(torch.cat([l.cpu() for _, l in train_set]), T), device=dev)
    # GAT and the pool read-out have no model-level plan: their graph mode is the captured layer-by-layer step
    # (eagcn_amd/graph_composed.py), which takes the dense batch
    composed = args.arch == 'GAT' or args.molfp == 'pool'
    graph = not args.no_graph and not (composed and args.loader == 'compact')
    if args.molfp == 'pool' and args.arch in ('Concate', 'Weighted_sum'):
        args.layers = 4                                        # the pool read-out needs layer 4's attention matrix
    model = EAGCN(n_bfeat, 24, *w1, *w2, d1, d2, T, args.dr, structure=args.arch, molfp_mode=args.molfp, n_layers=args.layers,
                  graph=graph, overlap_index=graph and not composed, validate='deferred' if graph else 'sync',
                  n_bucket=16 if (args.loader == 'compact' and graph) else 0).to(dev)
    model.apply(weights_init)
    if args.optimizer == 'flat' and not composed:
        from eagcn_amd.optim import FlatAdam
        opt = FlatAdam(model, lr=lr, weight_decay=wd)          # built after .to(dev) and weights_init: it re-homes the parameters
    else:
        opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=wd)
    t0 = time.perf_counter()
    for step in range(args.steps):
        b, labels = train_set[step % len(train_set)]
        if args.loader == 'compact':
            loss = training.train_step(model, opt, b[1:], labels, task, bce_w, bonds=b[0])
        else:
            loss = training.train_step(model, opt, b, labels, task, bce_w)
        if step % args.print_freq == 0 or step == args.steps - 1:
            metric = training.evaluate(model, val_set, task, T)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            shown = 'val mean AUC %.4f' % metric[1] if task == 'class' else 'val RMSE %.4f' % metric
            print('step %4d  loss %.5f  %s  (%.1f molecules/s incl. evaluation; %d captured runner(s))'
                  % (step, float(loss), shown, (step + 1) * args.batch / dt, len(model._runners)))


if __name__ == '__main__':
    main()
