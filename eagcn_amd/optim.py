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
        self.hyper = torch.tensor([lr, betas[0], betas[1], eps, weight_decay], dtype=torch.float32, device=dev)
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

    def _flat_grads(self):
        flat = self.model.flat_grad_buffer()
        if flat is not None and flat.numel() == self.flat.numel():
            return flat
        # autograd mode: the gradients are separate tensors -> one staging buffer of the same layout
        if self._gather is None:
            self._gather = torch.zeros_like(self.flat)
        views = self.plan.grad_views(self._gather)
        live = [(v, p.grad) for v, p in zip(views, self.plan.params) if p.requires_grad and p.grad is not None]
        self._gather.zero_()
        if live:
            torch._foreach_copy_([v for v, _ in live], [g for _, g in live])
        return self._gather

    def launch(self, flat_grad):
        """The update as one launch on the current stream (capturable)."""
        lib = L.load()
        L.check(lib.eagcn_adam_step(C.c_void_p(self.flat.data_ptr()), C.c_void_p(flat_grad.data_ptr()), C.c_void_p(self.exp_avg.data_ptr()),
                                    C.c_void_p(self.exp_avg_sq.data_ptr()), self.flat.numel(), C.c_void_p(self.hyper.data_ptr()),
                                    C.c_void_p(self.step_count.data_ptr()), C.c_void_p(self.ticket.data_ptr()),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'eagcn_adam_step')

    @torch.no_grad()
    def step(self):
        self.launch(self._flat_grads())

    def state_dict(self):
        return {'exp_avg': self.exp_avg.clone(), 'exp_avg_sq': self.exp_avg_sq.clone(), 'step': int(self.step_count),
                'hyper': self.hyper.clone()}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd['exp_avg'])
        self.exp_avg_sq.copy_(sd['exp_avg_sq'])
        self.step_count.fill_(int(sd['step']))
        self.hyper.copy_(sd['hyper'])
