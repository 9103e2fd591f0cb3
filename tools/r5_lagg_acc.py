"""edge-parameter gradients of 256-atom molecules: dense (agg.hip) and LDS-staged (lagg.hip) kernels against the float64 oracle"""
import os, sys, subprocess, tempfile, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
code = '''
import sys, torch
sys.path.insert(0, %r)
from eagcn_amd import EAGCN
from eagcn_amd.synthetic import make_batch
mb = make_batch(B=9, n_max=256, n_med=250, rel_channels=(28, 4, 2, 2, 2), seed=31, n_tasks=3)
dense = [t.cuda() for t in mb.dense()]
torch.manual_seed(3)
m = EAGCN(28, 24, *[16] * 5, *[40] * 5, 64, 32, 3, 0.0, structure='Weighted_sum', n_layers=2, grad_mode='direct').cuda().train()
with torch.no_grad():
    m.bn_den1.bias.fill_(6.0); m.bn_den2.bias.fill_(6.0)
torch.manual_seed(4)
cot = torch.randn(9, 3, device='cuda')
out, _, gr = m(*dense)
((out * cot).sum() + 0.1 * gr.sum()).backward()
torch.save({'sd': {k: v.cpu() for k, v in m.state_dict().items()}, 'cot': cot.cpu(), 'g': {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None}}, sys.argv[1])
''' % root
res = {}
with tempfile.TemporaryDirectory() as d:
    for agg in ('dense', 'lds'):
        path = os.path.join(d, agg + '.pt')
        r = subprocess.run([sys.executable, '-c', code, path], env=dict(os.environ, EAGCN_AGG=agg), capture_output=True, text=True, timeout=500)
        assert r.returncode == 0, r.stderr[-3000:]
        res[agg] = torch.load(path)
from eagcn_amd.synthetic import make_batch
from oracle.eagcn_ref import RefEAGCN
mb = make_batch(B=9, n_max=256, n_med=250, rel_channels=(28, 4, 2, 2, 2), seed=31, n_tasks=3)
cpu = [t.double() if t.is_floating_point() else t for t in mb.dense()]
ref = RefEAGCN(28, 24, [16] * 5, [40] * 5, 64, 32, 3, 0.0, n_layers=2, structure='Weighted_sum').double()
ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in res['dense']['sd'].items()}, strict=True)
ref.train()
out, _, gr = ref(*cpu)
((out * res['dense']['cot'].double()).sum() + 0.1 * gr.sum()).backward()
scale = max(p.grad.abs().max().item() for p in ref.parameters() if p.grad is not None)
for k, p in ref.named_parameters():
    if p.grad is None or not (k.endswith('self_r') or k.endswith('att.weight')): continue
    ed = (res['dense']['g'][k].double() - p.grad).abs().max().item(); el = (res['lds']['g'][k].double() - p.grad).abs().max().item()
    print('%-28s own max %.3e  |dense - f64| %.2e  |lds - f64| %.2e   (scale %.1f)' % (k, p.grad.abs().max().item(), ed, el, scale))
