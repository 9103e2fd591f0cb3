"""Parity at BASELINE.json's full shapes through size-independent properties (the CPU oracle would
take minutes there): permutation equivariance over the molecules of a batch, linearity of the
backward pass in the cotangent, model engine == layer-wise composition, graph replay == eager.
The HEAD's relus are held on their linear branch in these tests (BatchNorm shifts at +6, see _setup): they are exercised with live
gates against the oracle at batch 1024 / 1300 with narrow views (tests/test_gpu_parity.py::test_model_vs_oracle_tox21_shape)."""
import pytest
import torch

from helpers import assert_grad_close, rel_err

pytestmark = pytest.mark.gpu

CONFIGS = {
    # BASELINE configs[1]: Tox21 12-task, 2-layer 5-view Concate, batch 256, N_pad 132
    'tox21_c2': dict(structure='Concate', n_layers=2, w1=[80] * 5, w2=[140] * 5, dens=(256, 64), nclass=12,
                     n_bfeat=28, B=256, n_max=132, n_med=16),
    # configs[2]: HIV 2-layer Weighted_sum (every view 500 / 1250 wide), N_pad 222, batch 1024
    'hiv_c3': dict(structure='Weighted_sum', n_layers=2, w1=[100] * 5, w2=[250] * 5, dens=(512, 128), nclass=1,
                   n_bfeat=28, B=1024, n_max=222, n_med=23),
    # configs[3] shape: Lipophilicity 3-layer Concate, N_pad 115, the 512-molecule shard of one GPU
    'lipo_c4': dict(structure='Concate', n_layers=3, w1=[60] * 5, w2=[100] * 5, dens=(128, 64), nclass=1,
                    n_bfeat=18, B=512, n_max=115, n_med=27),
    # configs[4]: synthetic roofline stress, K = 8 views, channels [32,4,2,2,2,2,2,2], 64 / 128 per view, every molecule
    # has all N = 256 atoms, the 1024-molecule shard of one GPU
    'c5_synth': dict(structure='Concate', n_layers=2, w1=[64] * 8, w2=[128] * 8, dens=(256, 64), nclass=1,
                     n_bfeat=32, chans=[32, 4, 2, 2, 2, 2, 2, 2], B=1024, n_max=256, n_med=None, all_full=True),
}


def _setup(name, **kw):
    from eagcn_amd import EAGCN, weights_init
    from eagcn_amd.synthetic import make_batch
    c = CONFIGS[name]
    torch.manual_seed(0)
    chans = c.get('chans', [c['n_bfeat'], 4, 2, 2, 2])
    mb = make_batch(B=c['B'], n_max=c['n_max'], n_med=c['n_med'], rel_channels=chans, seed=77,
                    all_full=c.get('all_full', False))
    model = EAGCN(c['n_bfeat'], 24, n_den1=c['dens'][0], n_den2=c['dens'][1], nclass=c['nclass'], dropout=0.0,
                  widths1=c['w1'], widths2=c['w2'], rel_channels=chans, structure=c['structure'],
                  n_layers=c['n_layers'], **kw)
    model.apply(weights_init)
    # Two evaluations of the same model that differ in ROUNDING (molecules permuted, fused against composed read-out, torch's head
    # against head2.hip) are compared here at batch sizes where a single flipped relu of the HEAD shows: with 1024 x (n_den1 +
    # n_den2) pre-activations one of them sits within fp32 rounding of zero in almost every instance, and relu' of that unit
    # decides a 1 / B share of several gradients (measured: bn_den1.bias 9e-4, den1.weight 7e-4, layer1.ave.weight 1.5e-4 between two
    # bit-deterministic paths whose forward outputs agree to 9e-7).  The head's BatchNorm shifts are therefore moved to +6 sigma:
    # its relus stay on their linear branch, every other operation (and every layer relu: those kernels see identical inputs
    # in both evaluations) is exercised as it is; the head's relu gating is pinned by the oracle tests at small sizes.
    with torch.no_grad():
        model.bn_den1.bias.fill_(6.0)
        model.bn_den2.bias.fill_(6.0)
    return c, mb, model.cuda().train()


def _grads(model):
    return {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize('name', sorted(CONFIGS))
def test_permutation_equivariance_and_backward_linearity_head_relus_linearised(name):
    c, mb, model = _setup(name)
    dense = list(mb.dense('cuda'))              # built on the device (HIV / C5: 8 - 13 GB of dense collate tensors)
    B = c['B']
    g1 = torch.randn(B, c['nclass'], device='cuda')
    g2 = torch.randn(B, c['nclass'], device='cuda')

    def run(inputs, cot):
        model.zero_grad(set_to_none=True)
        out, _, gr = model(*inputs)
        (out * cot).sum().backward()
        return out.detach().clone(), _grads(model)

    out_a, ga = run(dense, g1)
    # --- molecules permuted: outputs permute, every parameter gradient is unchanged (BatchNorm statistics are
    #     permutation invariant; packing order changes, so sums are re-associated: fp32 tolerance)
    perm = torch.randperm(B, device='cuda')
    out_p, gp = run([t[perm] for t in dense], g1[perm])
    assert rel_err(out_p.cpu(), out_a[perm].cpu(), name + ' permuted') < 2e-5
    scale = max(v.abs().max().item() for v in ga.values())
    for k in ga:
        assert_grad_close(gp[k], ga[k].cpu(), scale, k, rtol=2e-4, floor=2e-5)
    # --- backward is linear in the cotangent: grad(2*g1 - 3*g2) == 2*grad(g1) - 3*grad(g2)
    _, gb = run(dense, g2)
    _, gc = run(dense, 2.0 * g1 - 3.0 * g2)
    for k in ga:
        want = 2.0 * ga[k] - 3.0 * gb[k]
        assert_grad_close(gc[k], want.cpu(), 5.0 * scale, k, rtol=2e-4, floor=2e-5)


@pytest.mark.parametrize('name', ['tox21_c2', 'hiv_c3', 'c5_synth'])
def test_engine_composition_and_graph_agree_at_full_size_head_relus_linearised(name):
    c, mb, a = _setup(name, grad_mode='direct')
    _, _, b = _setup(name, grad_mode='direct', graph=True)
    b.load_state_dict(a.state_dict())
    dense = list(mb.dense('cuda'))
    cot = torch.randn(c['B'], c['nclass'], device='cuda')
    res = []
    for model, fn in ((a, a.forward), (a, a.forward_composed), (b, b.forward), (b, b.forward)):
        for p in model.parameters():
            p.grad = None
        out, _, gr = fn(*dense)
        (out * cot).sum().backward()
        res.append((out.detach().clone(), _grads(model)))
        if fn == a.forward_composed:          # keep BatchNorm running statistics of a and b in step
            pass
    ref_out, ref_g = res[0]
    scale = max(v.abs().max().item() for v in ref_g.values())
    for i, (out, g) in enumerate(res[1:], 1):
        assert rel_err(out.cpu(), ref_out.cpu(), '%s run %d' % (name, i)) < 1e-5, i
        for k in ref_g:
            # HIP path against HIP path (no oracle finishes at these sizes): 1e-5 of the tensor's own largest entry plus a floor of
            # 1e-5 of the case's largest gradient for the re-associated sums of the two routes (fused / separate read-out, slab
            # order)
            assert_grad_close(g[k], ref_g[k].cpu(), scale, '%s (run %d)' % (k, i), rtol=1e-5, floor=1e-5)
