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
    def forward(ctx, kind, outputs, labels, weight, scale=None):
        if not outputs.is_cuda or outputs.dtype != torch.float32:
            raise L.EagcnHipError('fused loss needs fp32 device logits (no CPU path)')
        lib = L.load()
        x = outputs.contiguous()
        y = labels.to(device=x.device, dtype=torch.float32).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        # graph mode hands over the buffer its captured backward reads d(loss)/d(out) from (eagcn_amd/graph.py):
        # the gradient is written straight into it
        slot = getattr(outputs, '_eagcn_grad_slot', None)
        if slot is not None and (slot.shape != x.shape or slot.device != x.device or slot.dtype != x.dtype):
            slot = None
        dx = slot if slot is not None else torch.empty_like(x)
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
        if scale is not None:                       # data-parallel global normalisation (parallel.dp_loss_scale): a device
            dx.mul_(scale)                          # scalar, no host sync; the gradient buffer is scaled in place
            loss = loss * scale
        ctx.save_for_backward(dx)
        ctx.slot = slot
        return loss

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        if ctx.slot is not None:                    # dx IS the captured backward's gradient buffer: scale in place
            return None, dx.mul_(g), None, None, None
        return None, dx * g, None, None, None


class _StepLoss(torch.Tensor):
    """Loss of a graph-mode training step.  ``loss.backward()`` with default arguments launches the captured
    backward directly: d(loss)/d(logits) already sits in the buffer that graph reads (the loss kernel wrote it
    there), so autograd's ones-tensor and the scaling kernel between the two graphs disappear.  Any other use
    (an explicit ``gradient``, ``retain_graph``, arithmetic on the loss, ...) goes through autograd as usual."""

    def backward(self, gradient=None, retain_graph=None, create_graph=False, inputs=None):
        d = self.__dict__.pop('_eagcn_direct', None)
        if d is not None and gradient is None and not retain_graph and not create_graph and inputs is None:
            runner, generation = d
            if generation == runner.generation and torch.is_grad_enabled():
                runner.backward(runner.dout, None, generation)
                return None
        return super().backward(gradient, retain_graph, create_graph, inputs)


def _apply(kind, outputs, labels, weight, scale=None):
    loss = _FusedLoss.apply(kind, outputs, labels, weight, scale)
    step = getattr(outputs, '_eagcn_step', None)     # (runner, generation) of a graph-mode forward (eagcn_amd/graph.py)
    if step is not None and getattr(outputs, '_eagcn_grad_slot', None) is not None:
        loss = loss.as_subclass(_StepLoss)
        loss._eagcn_direct = step
    return loss


def fused_classification_loss(outputs, labels, bce_weight, dp_global_norm=False, group=None):
    """train.py:326-331 in one kernel; bce_weight: [T,2] tensor (utils.py:681-700 ``set_weight``).
    dp_global_norm=True (data parallel): the loss is normalised by the number of labelled entries of the GLOBAL batch
    instead of this rank's shard (eagcn_amd.parallel.dp_loss_scale: one 1-element all-reduce, no host sync), so that the
    averaged gradients equal the reference's on the concatenated batch."""
    if not isinstance(bce_weight, torch.Tensor):
        bce_weight = torch.tensor(bce_weight, dtype=torch.float32, device=outputs.device)
    scale = None
    if dp_global_norm:
        from .parallel import dp_loss_scale
        scale = dp_loss_scale(labels.to(outputs.device), group)
    return _apply('bce', outputs, labels, bce_weight, scale)


def fused_regression_loss(outputs, labels):
    """train.py:321-325 (MSELoss on flattened outputs) in one kernel."""
    return _apply('mse', outputs, labels, None)
