"""one layer, Weighted_sum 5 x W wide on a W-wide packed input: lds vs dense aggregation (two subprocesses), every output compared"""
import os, sys, subprocess, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys, torch
sys.path.insert(0, %r)
from eagcn_amd import _lib, ops
from eagcn_amd.layers import GraphConv_Layer
from eagcn_amd.synthetic import make_batch
W = int(sys.argv[2]); structure = sys.argv[3]
mb = make_batch(B=48, n_max=60, n_med=16, rel_channels=(28, 4, 2, 2, 2), seed=21)
dense = [t.cuda() for t in mb.dense()]
adj, afm, rels = dense[0], dense[1], dense[2:7]
torch.manual_seed(3)
lay = GraphConv_Layer(W, 28, *[W] * 5, dropout=0.0, structure=structure, rel_channels=[28, 4, 2, 2, 2]).cuda().train()
index = ops.BatchIndex(adj, rels)
il = ops.ColLayout.single(W)
torch.manual_seed(5)
x = torch.randn(index.T, il.ld, device='cuda', requires_grad=True)
xout, pad_row, ol = lay.forward_packed(index, x, il)
torch.manual_seed(6)
cot = torch.randn_like(xout)
(xout * cot).sum().backward()
res = {'xout': xout.detach().cpu(), 'dx': x.grad.cpu()}
for k, p in lay.named_parameters():
    if p.grad is not None: res['g.' + k] = p.grad.cpu()
torch.save(res, sys.argv[1])
''' % root
import torch
with tempfile.TemporaryDirectory() as d:
    for W, structure in ((240, 'Weighted_sum'), (48, 'Weighted_sum'), (256, 'Weighted_sum'), (144, 'Concate')):
        res = {}
        for agg in ('dense', 'lds'):
            path = os.path.join(d, '%s.pt' % agg)
            r = subprocess.run([sys.executable, '-c', code, path, str(W), structure], env=dict(os.environ, EAGCN_AGG=agg), capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr[-3000:]
            res[agg] = torch.load(path)
        worst = sorted([((res['lds'][k] - v).abs().max().item() / max(v.abs().max().item(), 1e-30), k) for k, v in res['dense'].items()], reverse=True)
        print(W, structure, [(round(a, 7), k) for a, k in worst[:8]])
