"""a-11 / f-3: the reference's training loop (train.py:289-334: Adam with weight decay on the weighted masked BCE) and its
evaluation (train.py:130-186) on the HIP engine in graph mode, tracked against the CPU oracle over 50 optimizer steps:
loss curve, parameters, BatchNorm running statistics, num_batches_tracked, validation AUC."""
import numpy as np
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('task', ['class', 'reg'])
def test_fifty_adam_steps_track_the_oracle(task):
    from sklearn import metrics
    from eagcn_amd import EAGCN, training
    from eagcn_amd.synthetic import make_batch
    from oracle.eagcn_ref import RefEAGCN, classification_loss, regression_loss, weights_init_
    T = 4 if task == 'class' else 1
    w1, w2 = [16, 12, 8, 8, 8], [24, 12, 12, 12, 12]
    torch.manual_seed(2)
    ref = RefEAGCN(9, 24, w1, w2, 32, 16, T, 0.0, n_layers=2)
    weights_init_(ref)
    hip = EAGCN(9, 24, *w1, *w2, 32, 16, T, 0.0, n_layers=2, graph=True).cuda().train()
    hip.load_state_dict(ref.state_dict(), strict=True)
    mbs = [make_batch(B=32, n_max=40, n_med=12, rel_channels=(9, 4, 2, 2, 2), seed=60 + i, n_tasks=T, task=task) for i in range(4)]
    cpu = [(mb.dense(), torch.from_numpy(mb.labels)) for mb in mbs]
    dev = [(tuple(t.cuda() for t in d), l.cuda()) for d, l in cpu]
    bw = training.set_weight(torch.cat([l for _, l in cpu]), T) if task == 'class' else None
    bw_dev = torch.tensor(bw, device='cuda') if bw else None
    opt_r = torch.optim.Adam(ref.parameters(), lr=5e-4, weight_decay=1e-4)
    opt_h = torch.optim.Adam(hip.parameters(), lr=5e-4, weight_decay=1e-4)
    loss_r, loss_h = [], []
    for step in range(50):
        d, l = cpu[step % 4]
        opt_r.zero_grad()
        out, _, _ = ref(*d)
        lr_ = classification_loss(out, l, bw) if task == 'class' else regression_loss(out, l)
        lr_.backward()
        opt_r.step()
        loss_r.append(float(lr_))
        dd, ll = dev[step % 4]
        loss_h.append(training.train_step(hip, opt_h, dd, ll, task, bw_dev))
    loss_h = [float(x) for x in loss_h]
    curve = max(abs(a - b) / max(abs(b), 1e-6) for a, b in zip(loss_h, loss_r))
    assert rel_err(torch.tensor(loss_h), torch.tensor(loss_r), 'loss curve (50 steps)') < 2e-4, curve
    assert loss_r[-1] < loss_r[0]                      # it trains
    sd_h, sd_r = hip.state_dict(), ref.state_dict()
    worst = 0.0
    for k, v in sd_r.items():
        if k.endswith('num_batches_tracked'):
            assert int(sd_h[k]) == int(v) == 50, k
            continue
        if k.endswith('batch_norm.weight') or k.endswith('batch_norm.bias'):
            continue                                   # the reference's unused parameters
        d = (sd_h[k].double().cpu() - v.double()).abs().max().item()
        worst = max(worst, d / max(v.abs().max().item(), 1e-3))
        assert d <= 5e-4 * max(v.abs().max().item(), 1e-3), (k, d)
    rel_err(torch.tensor([worst]), torch.tensor([0.0]), 'worst parameter / buffer drift after 50 steps')
    # evaluation (train.py:130-211)
    val = [make_batch(B=32, n_max=40, n_med=12, rel_channels=(9, 4, 2, 2, 2), seed=90 + i, n_tasks=T, task=task) for i in range(2)]
    got = training.evaluate(hip, [(tuple(t.cuda() for t in mb.dense()), torch.from_numpy(mb.labels).cuda()) for mb in val], task, T)
    ref.eval()
    with torch.no_grad():
        outs = torch.cat([ref(*mb.dense())[0] for mb in val])
    labels = torch.cat([torch.from_numpy(mb.labels) for mb in val])
    if task == 'class':
        probs = torch.sigmoid(outs).numpy()
        for j in range(T):
            m = (labels[:, j] != -1).numpy()
            fpr, tpr, _ = metrics.roc_curve(labels[:, j].numpy()[m].astype(int), probs[:, j][m], pos_label=1)
            assert abs(got[0][j] - metrics.auc(fpr, tpr)) < 2e-3, (j, got[0][j])
    else:
        want = float(np.sqrt(metrics.mean_squared_error(outs.numpy().ravel(), labels.numpy().ravel())))
        assert abs(got - want) < 1e-3 * max(want, 1.0), (got, want)
    assert hip.training
