import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eagcn_amd import EAGCN, training
from eagcn_amd.synthetic import make_batch
from oracle.eagcn_ref import RefEAGCN, classification_loss, regression_loss, weights_init_
task = os.environ.get('TASK', 'class'); T = 4 if task == 'class' else 1
w1, w2 = [16, 12, 8, 8, 8], [24, 12, 12, 12, 12]
torch.manual_seed(2)
ref = RefEAGCN(9, 24, w1, w2, 32, 16, T, 0.0, n_layers=2); weights_init_(ref)
ref64 = RefEAGCN(9, 24, w1, w2, 32, 16, T, 0.0, n_layers=2).double(); ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
hip = EAGCN(9, 24, *w1, *w2, 32, 16, T, 0.0, n_layers=2, graph=os.environ.get('GRAPH', '1') == '1').cuda().train()
hip.load_state_dict(ref.state_dict(), strict=True)
mbs = [make_batch(B=32, n_max=40, n_med=12, rel_channels=(9, 4, 2, 2, 2), seed=60 + i, n_tasks=T, task=task) for i in range(4)]
cpu = [(mb.dense(), torch.from_numpy(mb.labels)) for mb in mbs]
dev = [(tuple(t.cuda() for t in d), l.cuda()) for d, l in cpu]
bw = training.set_weight(torch.cat([l for _, l in cpu]), T) if task == 'class' else None
bw_dev = torch.tensor(bw, device='cuda') if bw else None
opts = [torch.optim.Adam(m.parameters(), lr=5e-4, weight_decay=1e-4) for m in (ref, ref64, hip)]
for step in range(50):
    d, l = cpu[step % 4]
    ls = []
    for m, o, dt in ((ref, opts[0], torch.float32), (ref64, opts[1], torch.float64)):
        o.zero_grad()
        out, _, _ = m(*[t.to(dt) if t.is_floating_point() else t for t in d])
        lo = classification_loss(out.float(), l, bw) if task == 'class' else regression_loss(out.float(), l)
        if dt == torch.float64:
            lo = (torch.nn.functional.mse_loss(out.view(-1), l.double().view(-1)) if task == 'reg' else lo)
        lo.backward(); o.step(); ls.append(float(lo))
    dd, ll = dev[step % 4]
    lh = float(training.train_step(hip, opts[2], dd, ll, task, bw_dev))
    if step in (0, 1, 2, 3, 5, 10, 20, 30, 40, 49):
        print('step %2d  ref32 %.7f  ref64 %.7f  hip %.7f   |32-64| %.1e  |hip-64| %.1e' % (step, ls[0], ls[1], lh, abs(ls[0] - ls[1]), abs(lh - ls[1])))
