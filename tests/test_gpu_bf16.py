"""The bf16-product modes (BASELINE.json configs[1] "bf16"): hidden-layer products with operands rounded to bf16, fp32 accumulation
-- eagcn_set_gemm_mode(4): ONE bf16 plane per operand written by the producers, the kernel of the default mode (csrc/gemm_bx3.hip);
eagcn_set_gemm_mode(2): round 2's consumer-rounded form (csrc/gemm_x6.h).  The error table of all modes against the fp32 oracle is
printed: mode 0 (fp32 MFMA) and mode 3 (bf16 x 3 planes, the default) are the PARITY paths and must agree with the oracle to 1e-5.  Not the parity path -- this test states and holds ITS tolerance against the fp32
oracle: operands carry 2^-9 relative rounding, a K-term dot product of O(1) terms carries ~2^-9 relative error, after
BatchNorm and two layers outputs agree to ~6e-3 of their scale and gradients to ~5e-2 of the gradient scale (measured,
printed in the parity report); the stated tolerances are 2e-2 and 1e-1."""
import pytest
import torch

from eagcn_amd import EAGCN, _lib
from eagcn_amd.synthetic import make_batch
from helpers import rel_err

pytestmark = pytest.mark.gpu


def test_bf16_product_mode_error_against_the_fp32_oracle():
    from oracle.eagcn_ref import RefEAGCN, regression_loss
    lib = _lib.load()
    torch.manual_seed(21)
    w1, w2 = [80] * 5, [140] * 5                       # the Tox21 widths: layer 2 has K = 400 (hidden products only)
    ref = RefEAGCN(28, 24, w1, w2, 64, 32, 1, 0.0, structure='Concate', n_layers=2)
    sd0 = {k: v.clone() for k, v in ref.state_dict().items()}
    mb = make_batch(B=128, n_max=40, n_med=14, rel_channels=(28, 4, 2, 2, 2), seed=5, n_tasks=1, task="reg")   # (a head BatchNorm over a dozen rows would amplify any perturbation)
    dense = mb.dense()
    labels = torch.from_numpy(mb.labels)
    ref.train()
    out_r, _, gr_r = ref(*dense)
    regression_loss(out_r, labels).backward()
    errs = {}
    for mode in (0, 3, 4, 2):
        old = lib.eagcn_set_gemm_mode(mode)
        try:
            m = EAGCN(28, 24, widths1=w1, widths2=w2, n_den1=64, n_den2=32, nclass=1, dropout=0.0, n_layers=2)
            m.load_state_dict(sd0, strict=True)
            m = m.cuda().train()
            out, _, gr = m(*[t.cuda() for t in dense])
            torch.nn.functional.mse_loss(out.view(-1), labels.cuda().view(-1)).backward()
        finally:
            lib.eagcn_set_gemm_mode(old)
        e_out = rel_err(out.detach().cpu(), out_r.detach(), 'out, gemm mode %d' % mode)
        e_gr = rel_err(gr.detach().cpu(), gr_r.detach(), 'graph_rep, gemm mode %d' % mode)
        got = dict(m.named_parameters())
        scale = max(p.grad.abs().max().item() for p in ref.parameters() if p.grad is not None)
        e_g = max(((got[k].grad.cpu() - p.grad).abs().max().item() / scale, k) for k, p in ref.named_parameters()
                  if p.grad is not None)
        rel_err(torch.tensor([e_g[0]]), torch.tensor([0.0]), 'worst gradient / gradient scale (%s), gemm mode %d' % (e_g[1], mode))
        errs[mode] = (e_out, e_gr, e_g[0])
    names = {0: 'fp32 MFMA', 3: 'bf16 x 3 planes (default)', 4: 'one bf16 plane', 2: 'consumer-rounded bf16'}
    print('\nerror against the fp32 oracle at the Tox21 widths (B = 128): out / graph_rep relative to their scale, worst gradient / gradient scale')
    for mode in (0, 3, 4, 2):
        print('  gemm mode %d  %-28s out %.1e   graph_rep %.1e   gradients %.1e' % ((mode, names[mode]) + errs[mode]))
    for mode in (0, 3):                                                            # the parity paths
        assert errs[mode][0] < 1e-5 and errs[mode][1] < 1e-5 and errs[mode][2] < 2e-5, (mode, errs[mode])
    for mode in (4, 2):                                                            # the bf16 modes' stated tolerance
        assert errs[mode][0] < 2e-2 and errs[mode][1] < 2e-2 and errs[mode][2] < 1e-1, (mode, errs[mode])
        assert errs[mode][0] > 1e-5, mode                                          # (and it really ran in bf16)
