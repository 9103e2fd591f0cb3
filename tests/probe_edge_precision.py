"""Probe (not a test): where does the d att.weight error of the Lipo full-width oracle case come from?
Prints HIP / fp32-oracle / fp64-oracle values of the layer-1 attention gradients."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eagcn_amd import EAGCN
from eagcn_amd.synthetic import make_batch
from oracle.eagcn_ref import RefEAGCN, weights_init_

c = dict(structure='Concate', n_layers=int(os.environ.get('NL', 3)), w1=[60] * 5, w2=[100] * 5, dens=(128, 64), nclass=1,
         chans=[18, 4, 2, 2, 2], B=12, n_max=115, n_med=27)
torch.manual_seed(17)
mb = make_batch(B=c['B'], n_max=c['n_max'], n_med=c['n_med'], rel_channels=c['chans'], seed=23)
kw = dict(structure=c['structure'], n_layers=c['n_layers'], rel_channels=c['chans'])
ref = RefEAGCN(c['chans'][0], 24, c['w1'], c['w2'], c['dens'][0], c['dens'][1], c['nclass'], 0.0, **kw)
weights_init_(ref)
ref64 = RefEAGCN(c['chans'][0], 24, c['w1'], c['w2'], c['dens'][0], c['dens'][1], c['nclass'], 0.0, **kw).double()
ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
hip = EAGCN(c['chans'][0], 24, n_den1=c['dens'][0], n_den2=c['dens'][1], nclass=c['nclass'], dropout=0.0,
            widths1=c['w1'], widths2=c['w2'], **kw).cuda().train()
hip.load_state_dict(ref.state_dict(), strict=True)
cpu = mb.dense()
gsel = torch.randn(c['B'], c['nclass'])
for m, inp in ((ref, cpu), (ref64, [t.double() if t.is_floating_point() else t for t in cpu])):
    out, _, gr = m(*inp)
    ((out * gsel.to(out.dtype)).sum() + 0.1 * gr.sum()).backward()
out_h, _, gr_h = hip(*[t.cuda() for t in cpu])
((out_h * gsel.cuda()).sum() + 0.1 * gr_h.sum()).backward()
p32, p64, ph = dict(ref.named_parameters()), dict(ref64.named_parameters()), dict(hip.named_parameters())
for k in p32:
    if p32[k].grad is None or not ('att.weight' in k or 'self_r' in k):
        continue
    g32, g64, gh = p32[k].grad.double().flatten(), p64[k].grad.flatten(), ph[k].grad.double().cpu().flatten()
    print('%-28s |g|max %.3e  err32 %.2e  errHIP %.2e   (rel own: %.1e / %.1e)' % (k, g64.abs().max(), (g32 - g64).abs().max(), (gh - g64).abs().max(),
          (g32 - g64).abs().max() / g64.abs().max(), (gh - g64).abs().max() / g64.abs().max()))

# relu-boundary check: elements of a layer output that are zero on one side and positive on the other
from eagcn_amd import ops
hip2 = EAGCN(c['chans'][0], 24, n_den1=c['dens'][0], n_den2=c['dens'][1], nclass=c['nclass'], dropout=0.0,
             widths1=c['w1'], widths2=c['w2'], **kw).cuda().train()
hip2.load_state_dict(ref.state_dict(), strict=True)
ref64b = RefEAGCN(c['chans'][0], 24, c['w1'], c['w2'], c['dens'][0], c['dens'][1], c['nclass'], 0.0, **kw).double()
ref64b.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
dev = [t.cuda() for t in cpu]
with torch.no_grad():
    index = ops.BatchIndex(dev[0], dev[2:-1])
    outs_h = [ops.unpack_rows(index, lay, x, None).cpu().double() for x, _, lay in hip2.forward_layers(index, dev[1])]
    inp64 = [t.double() if t.is_floating_point() else t for t in cpu]
    outs_r = ref64b.layer_outputs(inp64[0], inp64[1], *inp64[2:-1])
    ref32b = RefEAGCN(c['chans'][0], 24, c['w1'], c['w2'], c['dens'][0], c['dens'][1], c['nclass'], 0.0, **kw)
    ref32b.load_state_dict(ref.state_dict())
    outs_32 = [o.double() for o in ref32b.layer_outputs(cpu[0], cpu[1], *cpu[2:-1])]
for l, (h, r, r32) in enumerate(zip(outs_h, outs_r, outs_32)):
    flip_h = ((h > 0) != (r > 0))
    flip_32 = ((r32 > 0) != (r > 0))
    cols = flip_h.nonzero()[:, 2].tolist()
    print('layer %d: relu mask differs from fp64 in %d (HIP) / %d (fp32 oracle) of %d elements; HIP flipped columns %s; max |value| there %.2e; max|out| %.2e'
          % (l + 1, int(flip_h.sum()), int(flip_32.sum()), h.numel(), cols[:8], float(torch.maximum(h, r)[flip_h].max()) if flip_h.any() else 0.0, float(r.max())))
