"""a-11 / f-3: the reference's training loop (train.py:289-334: Adam with weight decay on the weighted masked BCE / MSE) and
its evaluation (train.py:130-211) on the HIP engine in graph mode, tracked against the CPU oracle over 50 optimizer steps:
loss curve, parameters, BatchNorm running statistics, num_batches_tracked, validation AUC / RMSE.

Training is a chaotic map: the fp32 CPU oracle itself leaves the fp64 trajectory by 1e-4 .. 2e-2 within 50 Adam steps
(tests/probe_train_drift.py), so the yardstick is the fp64 oracle and the bar is the reference's own fp32 drift: the HIP
trajectory may not be further from the exact one than 3x what the fp32 oracle is (plus 1e-5).  Measured: the HIP path
stays within 1e-6 of the fp64 trajectory for 30+ steps (fp64 BatchNorm sums) while the fp32 oracle is already 1e-3 off."""
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
    from oracle.eagcn_ref import RefEAGCN, weights_init_
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
    ref64 = RefEAGCN(9, 24, w1, w2, 32, 16, T, 0.0, n_layers=2).double()
    ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
    cpu64 = [([t.double() if t.is_floating_point() else t for t in d], l) for d, l in cpu]
    opt_r = torch.optim.Adam(ref.parameters(), lr=5e-4, weight_decay=1e-4)
    opt_x = torch.optim.Adam(ref64.parameters(), lr=5e-4, weight_decay=1e-4)
    opt_h = torch.optim.Adam(hip.parameters(), lr=5e-4, weight_decay=1e-4)

    def loss_of(out, l):
        if task == 'reg':
            return torch.nn.functional.mse_loss(out.view(-1), l.to(out.dtype).view(-1))
        w = torch.tensor(bw, dtype=out.dtype)
        wt = ((l == 1).to(out.dtype) * w[:, 0].view(1, -1) + (l == 0).to(out.dtype) * w[:, 1].view(1, -1)).view(-1)
        return torch.nn.functional.binary_cross_entropy_with_logits(out.view(-1), l.to(out.dtype).view(-1), weight=wt,
                                                                  reduction='sum') / ((l == 1).sum() + (l == 0).sum()).to(out.dtype)
    loss_r, loss_x, loss_h = [], [], []
    for step in range(50):
        for m, o, data, acc in ((ref, opt_r, cpu, loss_r), (ref64, opt_x, cpu64, loss_x)):
            d, l = data[step % 4]
            o.zero_grad()
            out, _, _ = m(*d)
            lo = loss_of(out, l)
            lo.backward()
            o.step()
            acc.append(float(lo.detach()))
        dd, ll = dev[step % 4]
        loss_h.append(training.train_step(hip, opt_h, dd, ll, task, bw_dev))
    loss_h = [float(x.detach()) for x in loss_h]
    drift32, worst = 0.0, 0.0
    for t in range(50):
        drift32 = max(drift32, abs(loss_r[t] - loss_x[t]))
        e = abs(loss_h[t] - loss_x[t])
        worst = max(worst, e / max(abs(loss_x[t]), 1e-6))
        assert e <= 3.0 * drift32 + 1e-5 * max(abs(loss_x[t]), 1.0), (t, loss_h[t], loss_x[t], drift32)
    rel_err(torch.tensor([worst]), torch.tensor([0.0]), 'loss curve vs fp64 oracle (fp32 oracle drift %.1e)' % drift32)
    assert loss_x[-1] < loss_x[0]                      # it trains
    sd_h, sd_r, sd_x = hip.state_dict(), ref.state_dict(), ref64.state_dict()
    pworst = 0.0
    for k, v in sd_x.items():
        if k.endswith('num_batches_tracked'):
            assert int(sd_h[k]) == int(v) == 50, k
            continue
        if k.endswith('batch_norm.weight') or k.endswith('batch_norm.bias'):
            continue                                   # the reference's unused parameters
        if k.endswith('graph_conv.bias') or k == 'Graph_BN.bias' or (k.endswith('running_mean') and 'block' in k):
            # a bias in front of a training-mode BatchNorm has an analytically ZERO gradient: what reaches Adam is rounding
            # noise, which Adam normalises to +-lr steps -- a random walk that no two arithmetics share (the bias itself is
            # cancelled by the BatchNorm; the per-view running_mean tracks mean + bias and walks with it)
            continue
        scale = max(v.abs().max().item(), 1e-3)
        d_h = (sd_h[k].double().cpu() - v).abs().max().item()
        d_r = (sd_r[k].double() - v).abs().max().item()
        pworst = max(pworst, d_h / scale)
        # (Adam moves a parameter by at most lr per step: 1 % of that travel is allowed on top of the fp32 oracle's own drift)
        assert d_h <= 3.0 * d_r + 1e-5 * scale + 0.01 * 5e-4 * 50, (k, d_h, d_r)
    rel_err(torch.tensor([pworst]), torch.tensor([0.0]), 'worst parameter / buffer distance to the fp64 oracle after 50 steps')
    # evaluation (train.py:130-211): the oracle takes the HIP-trained weights, so that what is compared is the
    # evaluation path and not 50 steps of training drift (an AUC moves by 1/(pos*neg) per swapped pair)
    ref = ref64.float()
    ref.load_state_dict({k: v.detach().cpu() for k, v in hip.state_dict().items()})
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
