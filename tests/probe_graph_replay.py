"""Probe: graph replay vs eager at widths that put the hidden-layer products on the wave-autonomous GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eagcn_amd import EAGCN, _lib
from eagcn_amd.synthetic import make_batch
from eagcn_amd.models import weights_init
lib = _lib.load()
name = os.environ.get('CASE', 'lipo')
c = {'lipo': dict(structure='Concate', n_layers=3, w1=[60] * 5, w2=[100] * 5, dens=(128, 64), nclass=1, chans=[18, 4, 2, 2, 2], B=12, n_max=115, n_med=27, all_full=False),
     'c5': dict(structure='Concate', n_layers=2, w1=[64] * 8, w2=[128] * 8, dens=(256, 64), nclass=1, chans=[32, 4, 2, 2, 2, 2, 2, 2], B=8, n_max=256, n_med=None, all_full=True)}[name]
torch.manual_seed(3)
mb = make_batch(B=c['B'], n_max=c['n_max'], n_med=c['n_med'], rel_channels=c['chans'], seed=23, all_full=c['all_full'])
kw = dict(structure=c['structure'], n_layers=c['n_layers'], rel_channels=c['chans'])
def mk(graph):
    torch.manual_seed(5)
    m = EAGCN(c['chans'][0], 24, n_den1=c['dens'][0], n_den2=c['dens'][1], nclass=c['nclass'], dropout=0.0, widths1=c['w1'], widths2=c['w2'], grad_mode='direct', graph=graph, **kw)
    m.apply(weights_init)
    return m.cuda().train()
a, b = mk(False), mk(True)
b.load_state_dict(a.state_dict())
sd0 = {k: v.clone() for k, v in a.state_dict().items()}
dev = [t.cuda() for t in mb.dense()]
gsel = torch.randn(c['B'], c['nclass'], device='cuda')
def run(m):
    m.load_state_dict(sd0)
    for p in m.parameters(): p.grad = None
    out, _, gr = m(*dev)
    ((out * gsel).sum() + 0.1 * gr.sum()).backward()
    torch.cuda.synchronize()
    return out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
o_ref, g_ref = run(a)
for rep in range(6):
    o, g = run(b)
    eo = ((o - o_ref).abs().max() / o_ref.abs().max()).item()
    worst = max(((g[k] - g_ref[k]).abs().max().item() / max(g_ref[k].abs().max().item(), 1e-30), k) for k in g_ref)
    nbad = sum(1 for k in g_ref if (g[k] - g_ref[k]).abs().max().item() > 1e-4 * max(g_ref[k].abs().max().item(), 1e-30))
    print('%s rep %d: out err %.2e, worst grad err %.2e (%s), %d/%d grads off, timeouts %d' % (name, rep, eo, worst[0], worst[1], nbad, len(g_ref), lib.eagcn_gemm_sk_timeouts()))
