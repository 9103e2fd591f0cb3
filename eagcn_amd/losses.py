"""Losses of the reference training loop (train.py:321-331).

``classification_loss`` / ``regression_loss`` are tensor-op restatements (any device);
``fused_classification_loss`` / ``fused_regression_loss`` call the HIP kernels of csrc/loss.hip:
loss value and d loss / d logits in ONE launch (the reference builds the class-weight tensor with a
B x T Python double loop, utils.py:653-679, and then runs ~10 small ops)."""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib as L


def class_weight_tensor(bce_weight, labels):
    """utils.py:653-679: label 1 -> w[j][0], label 0 -> w[j][1], anything else (missing) -> 0."""
    w = torch.as_tensor(bce_weight, dtype=torch.float32, device=labels.device)
    return ((labels == 1).to(torch.float32) * w[:, 0].view(1, -1) +
            (labels == 0).to(torch.float32) * w[:, 1].view(1, -1)).reshape(-1)


def classification_loss(outputs, labels, bce_weight):
    """train.py:326-331: weighted BCE-with-logits, summed, divided by the number of labels in {0,1}."""
    weights = class_weight_tensor(bce_weight, labels)
    non_nan = ((labels == 1).sum() + (labels == 0).sum()).to(torch.float32)
    return F.binary_cross_entropy_with_logits(outputs.view(-1), labels.float().view(-1), weight=weights,
                                              reduction='sum') / non_nan


def regression_loss(outputs, labels):
    """train.py:321-325."""
    return F.mse_loss(outputs.view(-1), labels.float().view(-1))


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, outputs, labels, weight):
        if not outputs.is_cuda or outputs.dtype != torch.float32:
            raise L.EagcnHipError('fused loss needs fp32 device logits (no CPU path)')
        lib = L.load()
        x = outputs.contiguous()
        y = labels.to(device=x.device, dtype=torch.float32).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if kind == 'bce':
            B, T = x.shape
            w = weight.to(device=x.device, dtype=torch.float32).contiguous()
            if tuple(w.shape) != (T, 2) or tuple(y.shape) != (B, T):
                raise L.EagcnHipError('bce loss: logits %s labels %s weight %s' % (tuple(x.shape), tuple(y.shape), tuple(w.shape)))
            L.check(lib.eagcn_bce_loss(x.data_ptr(), y.data_ptr(), w.data_ptr(), B, T, loss.data_ptr(),
                                       dx.data_ptr(), stream), 'eagcn_bce_loss')
        else:
            if y.numel() != x.numel():
                raise L.EagcnHipError('mse loss: %d predictions, %d targets' % (x.numel(), y.numel()))
            L.check(lib.eagcn_mse_loss(x.data_ptr(), y.data_ptr(), x.numel(), loss.data_ptr(), dx.data_ptr(), stream),
                    'eagcn_mse_loss')
        ctx.save_for_backward(dx)
        # graph mode hands over the buffer its captured backward reads d(loss)/d(out) from: scaling into it
        # saves the copy (eagcn_amd/graph.py)
        slot = getattr(outputs, '_eagcn_grad_slot', None)
        ctx.slot = slot if (slot is not None and slot.shape == x.shape and slot.device == x.device) else None
        return loss

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        if ctx.slot is not None:
            return None, torch.mul(dx, g, out=ctx.slot), None, None
        return None, dx * g, None, None


def fused_classification_loss(outputs, labels, bce_weight):
    """train.py:326-331 in one kernel; bce_weight: [T,2] tensor (utils.py:681-700 ``set_weight``)."""
    if not isinstance(bce_weight, torch.Tensor):
        bce_weight = torch.tensor(bce_weight, dtype=torch.float32, device=outputs.device)
    return _FusedLoss.apply('bce', outputs, labels, bce_weight)


def fused_regression_loss(outputs, labels):
    """train.py:321-325 (MSELoss on flattened outputs) in one kernel."""
    return _FusedLoss.apply('mse', outputs, labels, None)
