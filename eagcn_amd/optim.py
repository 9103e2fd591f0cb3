"""The optimizer step of the reference's training loop (train.py:303 ``optim.Adam(model.parameters(), lr, weight_decay=wd)``,
train.py:334 ``optimizer.step()``) as ONE kernel over ONE flat buffer.

``torch.optim.Adam`` walks ~100 parameter tensors per step (a ``foreach`` pass of a dozen launches).  ``FlatAdam`` re-homes every
hot-path parameter of an ``EAGCN`` into one contiguous fp32 buffer (``p.data`` become views of it, in the order and with the
16-byte alignment of the model's flat gradient buffer, ``ops.ModelPlan.offsets``), keeps the two moment estimates in buffers of
the same layout and updates all of it with ``eagcn_adam_step`` (csrc/loss.hip): same arithmetic as ``torch.optim.Adam`` (L2
``weight_decay`` added to the gradient, bias corrections in double precision).  Hyper-parameters and the step count live in
device memory, so the launch can be captured: ``EAGCN.fused_step(..., optimizer=opt)`` /
``training.train_step(model, opt, ...)`` put it INSIDE the step graph (forward + loss + backward [+ gradient all-reduce] +
update = one graph launch per training step).  Parameters that never receive a gradient (the reference leaves 48 of 177
without one: unused AFM_BatchNorm affine terms, ``self_r`` / ``ave_A`` of the layer wrappers) are not touched, as with
``torch.optim.Adam``, which skips ``grad is None``.
"""
import ctypes as C

import torch

from . import _lib as L


class FlatAdam:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if getattr(model, 'structure', None) == 'GAT' or getattr(model, 'molfp_mode', None) == 'pool':
            raise L.EagcnHipError('FlatAdam covers the models with a model-level plan (Concate / Weighted_sum / GCN, sum / ave read-out); '
                                  'use torch.optim.Adam for GAT / pool models')
        self.model = model
        plan = model.plan()
        self.plan = plan
        dev = plan.params[0].device
        if dev.type != 'cuda':
            raise L.EagcnHipError('FlatAdam needs the model on the GPU')
        n = plan.offsets[-1]
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, view in zip(plan.params, plan.grad_views(self.flat)):
                if not p.requires_grad:
                    continue                       # frozen entries of the plan (the constant attention weights of Vanilla_GCN): their
                                                   # slot stays 0 in every buffer, so the kernel's update of it is 0 -= 0
                view.copy_(p.data)
                p.data = view                      # the module's parameter now IS a slice of the flat buffer
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.hyper = torch.tensor([lr, betas[0], betas[1], eps, weight_decay], dtype=torch.float64, device=dev)   # (torch's scalars are doubles)
        self.step_count = torch.zeros((), dtype=torch.int64, device=dev)
        self.ticket = torch.zeros((), dtype=torch.int32, device=dev)
        self._gather = None
        self.defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.param_groups = [dict(self.defaults, params=[p for p in plan.params if p.requires_grad])]

    # ---- torch.optim.Optimizer surface the training loop uses -------------------------------------------------------------
    def zero_grad(self, set_to_none=True):
        for p in self.param_groups[0]['params']:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def set_lr(self, lr):
        self.hyper[0] = float(lr)                  # (a device write: a captured step graph picks it up at its next replay)
        self.param_groups[0]['lr'] = float(lr)

    def reset_ticket(self):
        """After an aborted launch / failed capture: the kernel's last-workgroup ticket starts from zero again (it resets itself at the end
        of every COMPLETED launch; left half counted, later launches would never advance the step count)."""
        self.ticket.zero_()

    def _check_homes(self):
        """The module's parameters must still BE slices of the flat buffer (a later ``model.to()`` / ``.float()`` / assignment to
        ``p.data`` re-homes them and would leave this optimizer updating memory nobody reads)."""
        lo = self.flat.data_ptr()
        hi = lo + 4 * self.flat.numel()
        for p in self.param_groups[0]['params']:
            if not (lo <= p.data_ptr() < hi):
                raise L.EagcnHipError('FlatAdam: a parameter of the model no longer lives in the optimizer\'s flat buffer (the model was moved '
                                      'or a parameter\'s .data was re-assigned after the optimizer was built): build a new FlatAdam')

    def _launch(self, off, n, flat_grad, advance):
        lib = L.load()
        L.check(lib.eagcn_adam_step(C.c_void_p(self.flat.data_ptr() + 4 * off), C.c_void_p(flat_grad.data_ptr() + 4 * off),
                                    C.c_void_p(self.exp_avg.data_ptr() + 4 * off), C.c_void_p(self.exp_avg_sq.data_ptr() + 4 * off), n,
                                    C.c_void_p(self.hyper.data_ptr()), C.c_void_p(self.step_count.data_ptr()),
                                    C.c_void_p(self.ticket.data_ptr()) if advance else C.c_void_p(0),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'eagcn_adam_step')

    def launch(self, flat_grad):
        """The update as one launch on the current stream (capturable): every slot of the flat buffer has a gradient (graph /
        direct mode: the hot path produces all of them in every backward)."""
        self._check_homes()
        self._launch(0, self.flat.numel(), flat_grad, True)

    @torch.no_grad()
    def step(self):
        flat = self.model.flat_grad_buffer()
        if flat is not None and flat.numel() == self.flat.numel():
            return self.launch(flat)
        # autograd mode: the gradients are separate tensors.  torch.optim.Adam SKIPS a parameter whose .grad is None (no weight decay,
        # no moment update): the parameters that have one are gathered into a staging buffer of the flat layout and updated range by
        # range (adjacent live parameters share a launch); only the last launch advances the step count
        self._check_homes()
        if self._gather is None:
            self._gather = torch.zeros_like(self.flat)
        views = self.plan.grad_views(self._gather)
        offs = self.plan.offsets
        live = [(i, v, p.grad) for i, (v, p) in enumerate(zip(views, self.plan.params)) if p.requires_grad and p.grad is not None]
        if not live:
            return
        torch._foreach_copy_([v for _, v, _ in live], [g for _, _, g in live])
        ranges = []
        for i, _, _ in live:
            if ranges and ranges[-1][1] == offs[i]:
                ranges[-1][1] = offs[i + 1]
            else:
                ranges.append([offs[i], offs[i + 1]])
        for j, (a, b) in enumerate(ranges):
            self._launch(a, b - a, self._gather, j == len(ranges) - 1)

    def state_dict(self):
        g = self.param_groups[0]
        return {'exp_avg': self.exp_avg.clone(), 'exp_avg_sq': self.exp_avg_sq.clone(), 'step': int(self.step_count),
                'hyper': self.hyper.clone(),
                # the hyper-parameters as the Python floats they were given as (what load_state_dict restores from)
                'param_group': {'lr': float(g['lr']), 'betas': tuple(float(b) for b in g['betas']), 'eps': float(g['eps']),
                                'weight_decay': float(g['weight_decay'])}}

    def load_state_dict(self, sd):
        """The hyper-parameters are rebuilt in DOUBLE precision from the saved Python floats; a checkpoint that only holds the device
        tensor (older rounds) is taken as it is when it is float64 -- a float32 one (round 4) holds fp32 IMAGES of the values
        (0.99900001287...: exactly the 1.7e-5 error in 1 - beta2 the double hypers exist to avoid), so its entries are rounded
        back to the shortest decimal that has that image.
        One step count for the whole buffer: torch.optim.Adam keeps a count per PARAMETER; with the range-by-range update of the
        autograd mode a parameter whose first gradient arrives in a later step gets the global count's bias correction here."""
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        self.step_count.fill_(int(sd['step']))
        if 'param_group' in sd:
            g = sd['param_group']
            vals = [g['lr'], g['betas'][0], g['betas'][1], g['eps'], g['weight_decay']]
        else:
            h = sd['hyper']
            vals = [float(v) for v in h.double().cpu()]
            if h.dtype == torch.float32:
                import numpy as np
                vals = [float(np.format_float_positional(np.float32(v), unique=True, trim='-')) if np.isfinite(v) else v for v in vals]
        self.hyper.copy_(torch.tensor(vals, dtype=torch.float64))
        self.param_groups[0].update(lr=vals[0], betas=(vals[1], vals[2]), eps=vals[3], weight_decay=vals[4])
