"""Data parallelism over the GPUs of one node: one process per GPU, the molecule batch sharded
across ranks, ONE sum all-reduce of a flat fp32 gradient buffer per step over RCCL/xGMI
(torch.distributed backend "nccl" is RCCL on ROCm), then a 1/world scale.

The reference has no multi-GPU code at all; this is the only collective the hot
path needs: BatchNorm runs on each rank's shard ("local-BN": every shard is exactly a reference run
at batch B/world), parameters are replicated (<= 15 MB).  Messages are 2-15 MB, i.e. latency-bound
on xGMI, so gradients travel as one flat buffer instead of one collective per tensor.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    force = os.environ.get('EAGCN_FORCE_DIST', '0') == '1'      # exercise the RCCL path on one GPU
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of n_items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradientAllReducer:
    """Averages the gradients of `params` across ranks with a single collective.

    Parameters whose ``.grad`` is None on every rank (the reference leaves 48 of 177 parameters
    without gradient: layer-level self_r / ave_A and the unused AFM_BatchNorm weight/bias) are
    skipped; the set must be the same on all ranks, which holds because it is structural.
    """

    def __init__(self, params, group=None, model=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.model = model
        self._flat = None

    def _model_flat_buffer(self):
        """grad_mode='direct' / graph mode keep every hot-path gradient in ONE flat buffer that the
        .grad tensors are views of: reduce it in place, no gather / scatter copies."""
        fn = getattr(self.model, 'flat_grad_buffer', None)
        flat = fn() if fn is not None else None
        if flat is None:
            return None
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
        for p in self.params:
            if p.grad is not None and not (lo <= p.grad.data_ptr() < hi):
                return None                       # some gradient lives elsewhere: use the generic path
        return flat

    def __call__(self):
        if not dist.is_initialized():
            return
        if dist.get_world_size(self.group) == 1 and os.environ.get('EAGCN_FORCE_DIST', '0') != '1':
            return
        flat = self._model_flat_buffer()
        if flat is not None:
            if flat.is_cuda:            # RCCL averages in the collective itself: no separate scaling launch
                dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                flat.div_(dist.get_world_size(self.group))
            return
        grads = [p.grad for p in self.params if p.grad is not None]
        if not grads:
            return
        n = sum(g.numel() for g in grads)
        if self._flat is None or self._flat.numel() != n or self._flat.device != grads[0].device:
            self._flat = torch.empty(n, dtype=grads[0].dtype, device=grads[0].device)
        views = []
        o = 0
        for g in grads:
            views.append(self._flat[o:o + g.numel()].view_as(g))
            o += g.numel()
        torch._foreach_copy_(views, grads)
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        self._flat.div_(dist.get_world_size(self.group))
        torch._foreach_copy_(grads, views)
