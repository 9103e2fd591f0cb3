"""Graph mode of the models without a model-level plan -- the GAT baseline and the Diff_Pooling read-out:
the layer-by-layer step (forward_composed + fused loss + autograd backward) captured as ONE HIP graph over static buffers
(eagcn_amd/graph_composed.py) must reproduce the eager step: same kernels, same order, the batch index capacity-sized
instead of exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = {
    'GAT-sum': dict(structure='GAT', molfp_mode='sum', n_layers=2),
    'GAT-pool': dict(structure='GAT', molfp_mode='pool', n_layers=2),
    'GCN-pool': dict(structure='GCN', molfp_mode='pool', n_layers=2),
    'Concate-pool': dict(structure='Concate', molfp_mode='pool', n_layers=4),
    'Weighted-pool': dict(structure='Weighted_sum', molfp_mode='pool', n_layers=4),
}
NCLASS = 3


def _models(case, dropout):
    from eagcn_amd import EAGCN
    torch.manual_seed(0)
    kw = dict(CASES[case])
    eager = EAGCN(9, 24, *[8] * 5, *[12] * 5, 32, 16, NCLASS, dropout, rel_channels=(9, 4, 2, 2, 2), **kw).cuda().train()
    graph = EAGCN(9, 24, *[8] * 5, *[12] * 5, 32, 16, NCLASS, dropout, rel_channels=(9, 4, 2, 2, 2), graph=True,
                  graph_outputs='copy', **kw).cuda().train()
    graph.load_state_dict(eager.state_dict())
    for m in (eager, graph):
        if m.structure == 'GAT' and dropout == 0.0:          # the reference hard-codes the attention dropout (layers.py:104)
            for layer in m.graph_layers():
                layer.graph_conv.dropout = 0.0
    return eager, graph


def _batch(seed, B=10, N=26):
    from eagcn_amd.synthetic import bce_weights, make_batch
    mb = make_batch(B=B, n_max=N, n_med=9, rel_channels=(9, 4, 2, 2, 2), seed=seed, n_tasks=NCLASS)
    dense = mb.dense(torch.device('cuda'))
    labels = torch.from_numpy(mb.labels).cuda()
    return dense, labels, torch.tensor(bce_weights(NCLASS), dtype=torch.float32, device='cuda')


def _close(a, b, what, tol=2e-6):
    scale = max(float(b.abs().max()), 1e-12)
    err = float((a - b).abs().max()) / scale
    assert err <= tol, (what, err)


@pytest.mark.parametrize('case', sorted(CASES))
def test_captured_step_equals_the_eager_step(case):
    from eagcn_amd.losses import fused_classification_loss
    eager, graph = _models(case, 0.0)
    for step, seed in enumerate((3, 4, 5, 6)):
        dense, labels, bw = _batch(seed)
        keep = step == 3                                  # last step: gradients are NOT cleared first -> accumulation
        for m in (eager, graph):
            if not keep:
                for p in m.parameters():
                    p.grad = None
        out_e, _, grep_e = eager(*dense)
        loss_e = fused_classification_loss(out_e, labels, bw)
        loss_e.backward()
        loss_g, (out_g, atom_g, grep_g) = graph.fused_step(dense, labels, 'class', bw)
        torch.cuda.synchronize()
        _close(loss_g, loss_e.detach(), '%s step %d loss' % (case, step))
        _close(out_g, out_e.detach(), '%s step %d out' % (case, step))
        _close(grep_g, grep_e.detach(), '%s step %d graph_rep' % (case, step))
        ge = dict(eager.named_parameters())
        for k, p in graph.named_parameters():
            if ge[k].grad is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
                continue
            assert p.grad is not None, k
            _close(p.grad, ge[k].grad, '%s step %d d %s' % (case, step, k), tol=5e-6)
        for (k, v), (_, w) in zip(graph.state_dict().items(), eager.state_dict().items()):
            if v.dtype.is_floating_point:
                _close(v, w, '%s step %d buffer %s' % (case, step, k))
            else:
                assert torch.equal(v, w), k
    runner = next(iter(graph._runners.values()))
    assert runner.replays == 4, runner.replays            # every batch is a replay of the recorded step


@pytest.mark.parametrize('case', ['GAT-sum', 'Concate-pool'])
def test_replays_draw_fresh_dropout_masks_and_eval_graph_equals_eager_eval(case):
    eager, graph = _models(case, 0.3)
    dense, labels, bw = _batch(11)
    outs = []
    for _ in range(4):
        for p in graph.parameters():
            p.grad = None
        loss, (out, _, _) = graph.fused_step(dense, labels, 'class', bw)
        assert torch.isfinite(loss).item() and all(torch.isfinite(p.grad).all().item() for p in graph.parameters() if p.grad is not None)
        outs.append(out.clone())
    assert not torch.equal(outs[1], outs[2]) and not torch.equal(outs[2], outs[3])     # same batch, different masks per replay
    # eval mode under no_grad: forward-only graph == eager eval forward (running statistics, no dropout)
    eager.load_state_dict(graph.state_dict())
    eager.eval(); graph.eval()
    with torch.no_grad():
        for seed in (21, 22, 23):
            d2, _, _ = _batch(seed)
            o_e, _, g_e = eager(*d2)
            o_g, a_g, g_g = graph(*d2)
            _close(o_g, o_e, '%s eval out' % case)
            _close(g_g, g_e, '%s eval graph_rep' % case)
    runner = next(iter(graph._runners.values()))
    assert runner.eval_graph is not None and runner.replays == 7


@pytest.mark.parametrize('case', ['GAT-sum', 'Concate-pool'])
def test_gradient_reducer_and_train_step_on_a_composed_graph_model(case):
    """The data-parallel helpers on a graph-mode GAT / pool model: its runners are ComposedRunners (no flat gradient buffer):
    flat_grad_buffer() answers None instead of raising, GradientAllReducer falls back to its generic path, and
    training.train_step with a reducer / the global loss normalisation takes the unfused path (the captured step of these
    models accepts neither) and yields the same gradients as the captured step."""
    from eagcn_amd.parallel import GradientAllReducer
    from eagcn_amd.training import train_step
    _, graph = _models(case, 0.0)
    dense, labels, bw = _batch(7)
    for p in graph.parameters():
        p.grad = None
    loss0, _ = graph.fused_step(dense, labels, 'class', bw)          # a ComposedRunner now sits in graph._runners
    want = {k: p.grad.clone() for k, p in graph.named_parameters() if p.grad is not None}
    assert graph.flat_grad_buffer() is None
    red = GradientAllReducer(graph.parameters(), model=graph)
    assert red._model_flat_buffer() is None
    red()                                                            # (no process group: a no-op, but it must not raise)

    class _NoStep:                                                   # the gradients are compared, not a parameter update
        def zero_grad(self, set_to_none=True):
            for p in graph.parameters():
                p.grad = None

        def step(self):
            pass
    sd = {k: v.clone() for k, v in graph.state_dict().items()}
    graph.load_state_dict(sd)
    loss1 = train_step(graph, _NoStep(), dense, labels, 'class', bw, dp_global_norm=True, reducer=red)
    _close(loss1.detach(), loss0.detach(), 'loss')
    for k, g in want.items():
        _close(dict(graph.named_parameters())[k].grad, g, k, tol=1e-5)
