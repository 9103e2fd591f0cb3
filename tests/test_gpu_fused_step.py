"""EAGCN.fused_step (forward + loss + backward as one captured graph) against the separate calls of the same model."""
import copy

import pytest
import torch

from eagcn_amd.losses import fused_classification_loss, fused_regression_loss
from eagcn_amd.models import EAGCN
from eagcn_amd.synthetic import bce_weights, make_batch
from helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('task,structure', [('class', 'Concate'), ('reg', 'Weighted_sum')])
def test_fused_step_matches_forward_loss_backward(task, structure):
    dev = torch.device('cuda', 0)
    T = 5 if task == 'class' else 1
    torch.manual_seed(5)
    a = EAGCN(28, 24, dropout=0.0, structure=structure, n_layers=2, widths1=[16] * 5, widths2=[32] * 5, n_den1=64, n_den2=32,
              nclass=T, graph=True, validate='deferred').to(dev).train()
    b = copy.deepcopy(a)
    bw = torch.tensor(bce_weights(T), dtype=torch.float32, device=dev)
    for step in range(5):                      # both slots, eager first use and replays; changing labels and batches
        mb = make_batch(B=24, n_max=30, n_med=11, rel_channels=(28, 4, 2, 2, 2), seed=40 + step, n_tasks=T, task=task)
        dense = mb.dense(dev)
        labels = torch.from_numpy(mb.labels).to(dev)
        for m in (a, b):
            for p in m.parameters():
                p.grad = None
        out, _, gr = a(*dense)
        loss = (fused_regression_loss(out, labels) if task == 'reg' else fused_classification_loss(out, labels, bw))
        loss.backward()
        loss_f, (out_f, _, gr_f) = b.fused_step(dense, labels, task, bw)
        rel_err(loss_f, loss, 'loss, step %d' % step)
        rel_err(out_f, out, 'out')
        rel_err(gr_f, gr, 'graph_representation')
        assert torch.equal(loss_f, loss) and torch.equal(out_f, out)
        for (n, p), q in zip(a.named_parameters(), b.parameters()):
            if p.grad is None:
                assert q.grad is None, n
                continue
            assert torch.equal(p.grad, q.grad), (step, n, (p.grad - q.grad).abs().max().item())
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


@pytest.mark.parametrize('task,T,B,dropout', [('class', 12, 300, 0.3), ('class', 3, 40, 0.0), ('reg', 1, 300, 0.3), ('reg', 8, 17, 0.0)])
def test_one_launch_head_matches_separate_launches(task, T, B, dropout):
    """fused_step runs the head's forward, the loss and the head's backward as ONE launch (csrc/head2.hip head_all_kernel: phases
    behind device-scope barriers); forward() + fused loss + backward() run the same stages as eight launches.  Same tile bodies:
    bit-identical outputs, loss and gradients -- with dropout (same seeds), at a batch that is not a multiple of 16 and large
    enough for row-chunked weight gradients (B > 256), with a single regression target (scalar loads in the last stage)."""
    dev = torch.device('cuda', 0)
    torch.manual_seed(11)
    a = EAGCN(28, 24, dropout=dropout, structure='Concate', n_layers=2, widths1=[16] * 5, widths2=[32] * 5, n_den1=96, n_den2=40,
              nclass=T, graph=True, validate='deferred').to(dev).train()
    b = copy.deepcopy(a)
    bw = torch.tensor(bce_weights(T), dtype=torch.float32, device=dev) if task == 'class' else None
    for step in range(4):
        mb = make_batch(B=B, n_max=26, n_med=10, rel_channels=(28, 4, 2, 2, 2), seed=90 + step, n_tasks=T, task=task)
        dense = mb.dense(dev)
        labels = torch.from_numpy(mb.labels).to(dev)
        for m in (a, b):
            for p in m.parameters():
                p.grad = None
        torch.manual_seed(500 + step)            # (the dropout seeds of a step are drawn from torch's generator)
        out, _, gr = a(*dense)
        loss = (fused_regression_loss(out, labels) if task == 'reg' else fused_classification_loss(out, labels, bw))
        loss.backward()
        torch.manual_seed(500 + step)
        loss_f, (out_f, _, gr_f) = b.fused_step(dense, labels, task, bw)
        assert torch.equal(out_f, out) and torch.equal(gr_f, gr), (step, (out_f - out).abs().max().item())
        assert abs(float(loss_f) - float(loss)) <= 2e-7 * abs(float(loss)), (step, float(loss_f), float(loss))
        for (n, p), q in zip(a.named_parameters(), b.parameters()):
            if p.grad is None:
                assert q.grad is None, n
                continue
            assert torch.equal(p.grad, q.grad), (step, n, (p.grad - q.grad).abs().max().item())
    sa, sb = a.state_dict(), b.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_pipelined_steps_equal_synchronous_steps():
    """Sixteen fused steps issued back to back WITHOUT a synchronisation -- the host runs ahead, the index build of batch k + 1 / k + 2
    runs on the side stream under step k, slots and ring entries are reused, the batch-ready flag and the start counter
    (eagcn_model.wait_flag / start_signal) order the two streams -- over four different batches taken round-robin: every step's
    loss and the last step's gradients equal those of the same steps issued one at a time with a device synchronisation after each
    (a side stream that started one step early would clear the index of the running step: it showed as a FASTER step, not as a
    failing test, until this one)."""
    dev = torch.device('cuda', 0)
    torch.manual_seed(21)
    a = EAGCN(28, 24, dropout=0.0, structure='Concate', n_layers=2, widths1=[32] * 5, widths2=[48] * 5, n_den1=64, n_den2=32,
              nclass=6, graph=True, validate='deferred').to(dev).train()
    b = copy.deepcopy(a)
    bw = torch.tensor(bce_weights(6), dtype=torch.float32, device=dev)
    batches = []
    for j in range(4):
        mb = make_batch(B=192, n_max=100, n_med=24, rel_channels=(28, 4, 2, 2, 2), seed=300 + j, n_tasks=6)
        batches.append((mb.dense(dev), torch.from_numpy(mb.labels).to(dev)))
    torch.cuda.synchronize()
    losses_a, losses_b = [], []
    for step in range(16):                      # (a): one at a time
        dense, labels = batches[step % 4]
        for p in a.parameters():
            p.grad = None
        losses_a.append(a.fused_step(dense, labels, 'class', bw)[0].clone())
        torch.cuda.synchronize()
    for step in range(16):                      # (b): pipelined
        dense, labels = batches[step % 4]
        for p in b.parameters():
            p.grad = None
        losses_b.append(b.fused_step(dense, labels, 'class', bw)[0].clone())
    torch.cuda.synchronize()
    for step, (la, lb) in enumerate(zip(losses_a, losses_b)):
        assert torch.equal(la, lb), (step, float(la), float(lb))
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        if p.grad is not None:
            assert torch.equal(p.grad, q.grad), (n, (p.grad - q.grad).abs().max().item())


def test_fused_step_scale_and_accumulation():
    dev = torch.device('cuda', 0)
    torch.manual_seed(6)
    a = EAGCN(28, 24, dropout=0.0, structure='Concate', n_layers=2, widths1=[16] * 5, widths2=[32] * 5, n_den1=64, n_den2=32,
              nclass=3, graph=True).to(dev).train()
    b = copy.deepcopy(a)
    bw = torch.tensor(bce_weights(3), dtype=torch.float32, device=dev)
    mb = make_batch(B=16, n_max=24, n_med=10, rel_channels=(28, 4, 2, 2, 2), seed=77, n_tasks=3)
    dense = mb.dense(dev)
    labels = torch.from_numpy(mb.labels).to(dev)
    scale = torch.tensor(0.625, device=dev)
    for rep in range(3):                       # gradients stay attached: the second and third step ACCUMULATE
        out, _, _ = a(*dense)
        l1 = fused_classification_loss(out, labels, bw) * scale
        l1.backward()
        l2, _ = b.fused_step(dense, labels, 'class', bw, scale)
        assert abs(float(l1) - float(l2)) <= 1e-6 * abs(float(l1))
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        if p.grad is not None:
            d = (p.grad - q.grad).abs().max().item()
            assert d <= 2e-6 * max(p.grad.abs().max().item(), 1e-6), (n, d)


def test_bce_loss_large_batch_multi_workgroup_kernel():
    """B*T beyond 4096 takes the multi-workgroup loss kernel: value and gradient against the tensor-op restatement of
    train.py:326-331 (eagcn_amd.losses.classification_loss), twice in a row (the accumulator re-arms itself)."""
    from eagcn_amd.losses import classification_loss
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(3)
    B, T = 1024, 12
    x = (torch.randn(B, T, generator=g) * 3).to(dev)
    y = torch.randint(-1, 2, (B, T), generator=g).float().to(dev)
    w = torch.tensor(bce_weights(T), dtype=torch.float32, device=dev)
    for _ in range(2):
        xa = x.clone().requires_grad_(True)
        la = fused_classification_loss(xa, y, w)
        la.backward()
        xb = x.clone().requires_grad_(True)
        lb = classification_loss(xb, y, w)
        lb.backward()
        assert abs(float(la) - float(lb)) <= 2e-6 * abs(float(lb)), (float(la), float(lb))
        assert (xa.grad - xb.grad).abs().max().item() <= 2e-6 * xb.grad.abs().max().item()


def test_gemm_handoff_failure_is_loud():
    """A timed-out stream-K hand-off (csrc/gemm3.hip) sets a sticky host-visible word: the eager engine's next C call and the
    graph-replay loop (which makes no C call per step) both raise instead of continuing with a poisoned step."""
    from eagcn_amd import _lib as L
    lib = L.load()
    dev = torch.device('cuda', 0)
    torch.manual_seed(1)
    mb = make_batch(B=8, n_max=20, n_med=9, rel_channels=(28, 4, 2, 2, 2), seed=3, n_tasks=1, task='reg')
    dense = mb.dense(dev)
    labels = torch.from_numpy(mb.labels).to(dev)
    kw = dict(dropout=0.0, n_layers=2, widths1=[32] * 5, widths2=[16] * 5, n_den1=32, n_den2=16, nclass=1)
    g = EAGCN(28, 24, graph=True, validate='deferred', **kw).to(dev).train()
    e = EAGCN(28, 24, **kw).to(dev).train()
    for _ in range(3):
        g.fused_step(dense, labels, 'reg')
    e(*dense)
    torch.cuda.synchronize()
    assert lib.eagcn_gemm_sk_failed() == 0
    lib.eagcn_gemm_sk_inject_failure()
    try:
        with pytest.raises(L.EagcnHipError, match='hand-off'):
            g.fused_step(dense, labels, 'reg')
        with pytest.raises(L.EagcnHipError, match='hand-off'):
            e(*dense)
    finally:
        lib.eagcn_gemm_sk_reset_failed()
    g.fused_step(dense, labels, 'reg')            # usable again after the reset
    torch.cuda.synchronize()
    assert lib.eagcn_gemm_sk_failed() == 0
