import os, sys, subprocess, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys, torch
sys.path.insert(0, %r)
from eagcn_amd import EAGCN, _lib
from eagcn_amd.synthetic import make_batch
structure, nl, w1, w2 = sys.argv[1], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
mb = make_batch(B=48, n_max=60, n_med=16, rel_channels=(28, 4, 2, 2, 2), seed=21)
dense = [t.cuda() for t in mb.dense()]
torch.manual_seed(3)
m = EAGCN(28, 24, *[w1] * 5, *[w2] * 5, 64, 32, 3, 0.0, structure=structure, n_layers=nl, grad_mode='direct').cuda().train()
torch.manual_seed(4)
cot = torch.randn(48, 3, device='cuda')
out, _, gr = m(*dense)
((out * cot).sum() + 0.1 * gr.sum()).backward()
torch.cuda.synchronize()
sd = {k: v.detach().cpu() for k, v in m.state_dict().items() if 'running' in k}
torch.save({'gr': gr.detach().cpu(), 'out': out.detach().cpu(), 'g': {k: p.grad.cpu() for k, p in m.named_parameters() if p.grad is not None}, 'sd': sd}, sys.argv[2])
''' % root
import torch
EXTRA = {}
if len(sys.argv) > 1: EXTRA = dict(kv.split('=') for kv in sys.argv[1:])
with tempfile.TemporaryDirectory() as d:
    for cfg in [('Weighted_sum', 3, 48, 64)]:
        res = {}
        for agg, parts in (('dense', '3'), ('lds', '1')):
            path = os.path.join(d, 'x.pt')
            r = subprocess.run([sys.executable, '-c', code, cfg[0], path, str(cfg[1]), str(cfg[2]), str(cfg[3])], env=dict(os.environ, EAGCN_AGG=agg, EAGCN_LAGG_PARTS=parts, **EXTRA), capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-3000:]
            res[agg] = torch.load(path)
        a, ref = res['lds'], res['dense']
        scale = max(v.abs().max().item() for v in ref['g'].values())
        worst = sorted([((a['g'][k] - v).abs().max().item() / (1e-5 * v.abs().max().item() + 2e-6 * scale), k) for k, v in ref['g'].items()], reverse=True)
        print('gr', ((a['gr'] - ref['gr']).abs().max() / ref['gr'].abs().max()).item())
        print([(round(x, 2), k) for x, k in worst if not k.startswith('layer')])
        print([(round(x, 2), k) for x, k in worst if k.startswith('layer3')][:12])
        print(cfg, 'out', ((a['out'] - ref['out']).abs().max() / ref['out'].abs().max()).item(), 'worst d / tol', [(round(x, 2), k) for x, k in worst[:4]], flush=True)
