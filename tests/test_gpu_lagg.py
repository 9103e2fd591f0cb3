"""The LDS-staged bond-list aggregation (csrc/lagg.hip; reference layers.py:82-92 forward, and its autograd: transposed aggregation
+ the edge gradients of att.weight / self_r): the same operator as the matrix-core kernels of csrc/agg.hip, exact including the 1e-9
filler.  Since round 6 it is the DEFAULT in both directions for every layer whose padded size is at most 256 atoms (csrc/lagg.hip
lagg_use; Weighted_sum layers re-form dH from the upstream gradient in its staging); here it is compared with the dense kernels
(EAGCN_AGG=lds against EAGCN_AGG=dense, read once per process: subprocesses) over small / ragged / isolated-atom / self-loop batches,
both merges, widths that end in a half chunk, eager and graph mode, one and several column chunks per workgroup -- and with the CPU oracle."""
import os
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = '''
import sys, torch
sys.path.insert(0, %r)
from eagcn_amd import EAGCN
from eagcn_amd.synthetic import make_batch
structure, path, B, n_max, n_med, w1, w2, graph, iso = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]), float(sys.argv[9])
mb = make_batch(B=B, n_max=n_max, n_med=n_med, rel_channels=(28, 4, 2, 2, 2), seed=31, isolated_frac=iso, n_tasks=3)
dense = [t.cuda() for t in mb.dense()]
torch.manual_seed(3)
m = EAGCN(28, 24, *[w1] * 5, *[w2] * 5, 64, 32, 3, 0.0, structure=structure, n_layers=2, grad_mode='direct', graph=bool(graph)).cuda().train()
with torch.no_grad():                     # (head relus never gate: see tests/test_gpu_bx3.py)
    m.bn_den1.bias.fill_(6.0); m.bn_den2.bias.fill_(6.0)
torch.manual_seed(4)
cot = torch.randn(B, 3, device='cuda')
for it in range(3 if graph else 1):      # graph mode: eager step, capture, replay
    for p in m.parameters(): p.grad = None
    out, _, gr = m(*dense)
    ((out * cot).sum() + 0.1 * gr.sum()).backward()
torch.cuda.synchronize()
torch.save({'out': out.detach().cpu(), 'g': {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None}}, path)
''' % ROOT


def _run(agg, args, d, tag='', **env):
    path = os.path.join(d, agg + tag + '.pt')
    r = subprocess.run([sys.executable, '-c', CODE, args[0], path] + [str(a) for a in args[1:]], env=dict(os.environ, EAGCN_AGG=agg, **env),
                       capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stderr[-3000:]
    return torch.load(path)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('args', [('Concate', 24, 70, 25, 48, 64, 0, 0.05), ('Weighted_sum', 24, 70, 25, 48, 64, 0, 0.05),
                                  ('Concate', 40, 200, 60, 144, 80, 0, 0.0),
                                  ('Concate', 64, 33, 9, 80, 140, 1, 0.02), ('Weighted_sum', 300, 30, 6, 16, 16, 0, 0.1)],
                         ids=['concate-ragged', 'weighted-ragged', 'concate-200-atoms-half-chunks', 'concate-graph-mode',
                              'weighted-many-small-molecules'])
def test_lds_staged_aggregation_matches_the_dense_kernels(args):
    with tempfile.TemporaryDirectory() as d:
        a, ref = _run('lds', args, d), _run('dense', args, d)
    assert ((a['out'] - ref['out']).abs().max() / ref['out'].abs().max()).item() < 1e-5
    scale = max(v.abs().max().item() for v in ref['g'].values())
    worst = (0.0, '')
    for k, v in ref['g'].items():
        dd = (a['g'][k] - v).abs().max().item()
        # the edge parameters' gradients are sums of s_i (<dY'_i, P_j> - <dY'_i, Y'_i>) over all bonds / atoms: differences of fp32 dot
        # products, taken in a different order by the two kernels (per 32-column chunk here) -- both sides carry that rounding: the
        # tolerance the oracle tests give one side (tests/test_gpu_parity.py) x 1, not x 1/2
        edge = k.endswith('self_r') or k.endswith('att.weight')
        tol = (2e-5 if edge else 1e-5) * v.abs().max().item() + (4e-6 if edge else 2e-6) * scale
        worst = max(worst, (dd / tol, k))
        assert dd <= tol, (k, dd, v.abs().max().item(), scale)
    print('worst gradient distance / tolerance %.2f (%s)' % worst)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('args', [('Concate', 40, 200, 60, 144, 80, 0, 0.0), ('Weighted_sum', 24, 70, 25, 48, 64, 0, 0.05)],
                         ids=['concate-half-chunks', 'weighted-ragged'])
def test_several_column_chunks_per_workgroup_change_no_bit(args):
    """Forward launches with thousands of workgroups let one workgroup take several 32-column chunks of a block (its lists and row
    records are built once; csrc/lagg.hip lagg_grid): the same arithmetic per chunk, so the outputs are bit-identical to
    one chunk per workgroup.  Forced here (EAGCN_LAGG_CPW) at widths that end in a half chunk and in a partial group."""
    with tempfile.TemporaryDirectory() as d:
        one = _run('lds', args, d, tag='1', EAGCN_LAGG_CPW='1')
        for cpw in ('2', '3'):
            many = _run('lds', args, d, tag=cpw, EAGCN_LAGG_CPW=cpw, EAGCN_LAGG_CPW_BWD='1')
            assert torch.equal(one['out'], many['out'])
            scale = max(v.abs().max().item() for v in one['g'].values())
            for k, v in one['g'].items():          # (the backward is the same kernel on bit-identical inputs; its fp64 atomics are unordered)
                assert (v - many['g'][k]).abs().max().item() <= 1e-6 * scale, (cpw, k)


@pytest.mark.timeout(900)
def test_lds_staged_aggregation_vs_oracle_with_self_loops_and_isolated_atoms():
    """Against the CPU oracle (float64), on a batch with self bonds (the diagonal of the edge gradients is counted once) and atoms
    without bonds inside the stored rows (masked rows)."""
    code = '''
import sys, torch
sys.path.insert(0, %r)
from eagcn_amd import EAGCN
from eagcn_amd.synthetic import make_batch
from oracle.eagcn_ref import RefEAGCN, weights_init_      # checker only
mb = make_batch(B=20, n_max=90, n_med=30, rel_channels=(9, 4, 2, 2, 2), seed=5, isolated_frac=0.08, n_tasks=3)
cpu = list(mb.dense())
adj = cpu[0]
for b in range(0, 20, 3):                # self bonds on a few atoms that have bonds (type 0 in every view)
    i = int(adj[b].sum(1).argmax())
    adj[b, i, i] = 1.0
    for r in cpu[2:7]: r[b, 0, i, i] = 1.0
w1, w2 = [24, 16, 16, 16, 16], [40, 24, 24, 24, 24]
torch.manual_seed(0)
ref = RefEAGCN(9, 24, w1, w2, 32, 16, 3, 0.0, n_layers=2)
weights_init_(ref)
ref = ref.double()
hip = EAGCN(9, 24, *w1, *w2, 32, 16, 3, 0.0, n_layers=2, grad_mode='direct').cuda().train()
hip.load_state_dict({k: v.float() for k, v in ref.state_dict().items()}, strict=True)
torch.manual_seed(4)
cot = torch.randn(20, 3)
out_r, _, gr_r = ref(*[t.double() if t.is_floating_point() else t for t in cpu])
((out_r * cot.double()).sum() + 0.1 * gr_r.sum()).backward()
out_h, _, gr_h = hip(*[t.cuda() for t in cpu])
((out_h * cot.cuda()).sum() + 0.1 * gr_h.sum()).backward()
err = ((out_h.detach().cpu().double() - out_r.detach()).abs().max() / out_r.detach().abs().max()).item()
assert err < 1e-5, err
scale = max(p.grad.abs().max().item() for p in ref.parameters() if p.grad is not None)
worst = (0.0, '')
for (k, p), (_, q) in zip(ref.named_parameters(), hip.named_parameters()):
    if p.grad is None: continue
    dd = (q.grad.detach().cpu().double() - p.grad).abs().max().item()
    tol = 1e-5 * p.grad.abs().max().item() + 2e-6 * scale      # (north-star 1e-5; round 6: was 2e-5 + 4e-6, achieved 0.09 of that)
    worst = max(worst, (dd / tol, k))
    assert dd <= tol, (k, dd, tol)
print('LAGG_ORACLE_OK out %%.1e worst grad / tol %%.2f (%%s)' %% (err, worst[0], worst[1]))
''' % ROOT
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, EAGCN_AGG='lds'), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'LAGG_ORACLE_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
    print(r.stdout.strip().splitlines()[-1])


@pytest.mark.timeout(900)
def test_lds_staged_aggregation_at_256_atoms_vs_float64_oracle():
    """Molecules of ~250 atoms (the shape the path is taken for by default) against the float64 oracle.  The dense kernels are NOT the
    yardstick here: their edge-parameter gradients carry the cancellation of two full-width fp32 dot products per bond
    (|dense - f64| up to 6e-4 of a 0.5 gradient in this case, tools/r5_lagg_acc.py), the chunked partial differences of lagg.hip stay
    at 1e-6."""
    sys.path.insert(0, ROOT)
    code = '''
import sys, torch
sys.path.insert(0, %r)
from eagcn_amd import EAGCN
from eagcn_amd.synthetic import make_batch
mb = make_batch(B=9, n_max=256, n_med=250, rel_channels=(28, 4, 2, 2, 2), seed=31, n_tasks=3)
dense = [t.cuda() for t in mb.dense()]
torch.manual_seed(3)
m = EAGCN(28, 24, *[16] * 5, *[40] * 5, 64, 32, 3, 0.0, structure=sys.argv[2], n_layers=2, grad_mode='direct').cuda().train()
with torch.no_grad():
    m.bn_den1.bias.fill_(6.0); m.bn_den2.bias.fill_(6.0)
torch.manual_seed(4)
cot = torch.randn(9, 3, device='cuda')
out, _, gr = m(*dense)
((out * cot).sum() + 0.1 * gr.sum()).backward()
torch.save({'sd': {k: v.cpu() for k, v in m.state_dict().items()}, 'cot': cot.cpu(), 'out': out.detach().cpu(),
            'g': {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None}}, sys.argv[1])
''' % ROOT
    from eagcn_amd.synthetic import make_batch
    from oracle.eagcn_ref import RefEAGCN      # checker only
    for structure in ('Weighted_sum', 'Concate'):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, 'lds.pt')
            r = subprocess.run([sys.executable, '-c', code, path, structure], env=dict(os.environ, EAGCN_AGG='lds'), capture_output=True, text=True, timeout=500)
            assert r.returncode == 0, r.stderr[-3000:]
            got = torch.load(path)
        mb = make_batch(B=9, n_max=256, n_med=250, rel_channels=(28, 4, 2, 2, 2), seed=31, n_tasks=3)
        cpu = [t.double() if t.is_floating_point() else t for t in mb.dense()]
        ref = RefEAGCN(28, 24, [16] * 5, [40] * 5, 64, 32, 3, 0.0, n_layers=2, structure=structure).double()
        ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in got['sd'].items()}, strict=True)
        ref.train()
        out, _, gr = ref(*cpu)
        ((out * got['cot'].double()).sum() + 0.1 * gr.sum()).backward()
        assert ((got['out'].double() - out.detach()).abs().max() / out.detach().abs().max()).item() < 1e-5
        scale = max(p.grad.abs().max().item() for p in ref.parameters() if p.grad is not None)
        worst = (0.0, '')
        for k, p in ref.named_parameters():
            if p.grad is None:
                continue
            dd = (got['g'][k].double() - p.grad).abs().max().item()
            tol = 1e-5 * p.grad.abs().max().item() + 2e-6 * scale      # (round 6: was 2e-5 + 4e-6, achieved 0.08-0.11 of that)
            worst = max(worst, (dd / tol, k))
            assert dd <= tol, (structure, k, dd, tol)
        print('%s: worst gradient distance to the float64 oracle / tolerance %.2f (%s)' % (structure, worst[0], worst[1]))
