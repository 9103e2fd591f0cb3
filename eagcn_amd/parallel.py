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


def dp_loss_scale(labels, group=None):
    """Per-rank factor c_r = world * n_r / sum_r n_r (a 0-dim tensor on the labels' device, no host sync) that turns
    the reference's per-batch normalisation of the classification loss -- weighted BCE summed and divided by the number
    of labels in {0,1} (train.py:328-331) -- into the GLOBAL-batch normalisation under data parallelism:
        L_global = sum_r S_r / sum_r n_r       (S_r: this rank's weighted BCE sum, n_r: its labelled entries)
                 = mean_r ( c_r * S_r / n_r ) = mean_r ( c_r * L_r ).
    Scaling the local loss (and hence d loss / d logits) by c_r and AVERAGING the gradients over ranks (what
    GradientAllReducer does) therefore yields exactly the gradient of the loss the reference would compute on the
    concatenated batch; without it a shard with few labelled entries is over-weighted.  One 1-element all-reduce."""
    n = ((labels == 1) | (labels == 0)).sum().to(torch.float32).reshape(1)
    if not dist.is_initialized():
        return torch.ones((), dtype=torch.float32, device=labels.device)
    tot = n.clone()
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    return (n * float(dist.get_world_size(group)) / tot).reshape(())


# ---- sync-BatchNorm: the cross-rank part of a BatchNorm over the GLOBAL batch (SURVEY.md 8e "BN modes") ---------------
# Every rank reduces its own rows to per-channel partial sums (that is what the layer kernels produce anyway: the
# per-workgroup (sum y, sum y^2) slabs of agg.hip / (sum dH, sum dH*xhat) slabs of bn_bwd_reduce); the functions below are
# the two tiny collectives per layer and direction that turn them into statistics of the concatenated batch.
def sync_bn_forward_stats(sum_y, sum_y2, rows, group=None):
    """(sum y, sum y^2) per channel [F] (float64) and this rank's row count (B_r * N_pad_r, padded rows included, reference
    layers.py:408-412) -> (mean, biased var, unbiased var, total rows) of the global batch.  ONE all-reduce of 2F+1 doubles."""
    buf = torch.cat([sum_y.double().reshape(-1), sum_y2.double().reshape(-1),
                     torch.tensor([float(rows)], dtype=torch.float64, device=sum_y.device)])
    if dist.is_initialized():
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    F = sum_y.numel()
    M = buf[-1]
    mean = buf[:F] / M
    var = (buf[F:2 * F] / M - mean * mean).clamp_min(0.0)
    return mean, var, var * (M / (M - 1.0)), M


def sync_bn_backward_stats(sum_dh, sum_dh_xhat, group=None):
    """Per-channel (sum dH, sum dH * xhat) of this rank -> the same sums over the global batch (the c1, c2 terms of
    dY = scale * (dH - mean(dH) - xhat * mean(dH * xhat)) are means over ALL rows).  d gamma / d beta stay the LOCAL
    sums: like every other parameter gradient they are averaged by the gradient all-reduce.  ONE all-reduce of 2F doubles."""
    buf = torch.cat([sum_dh.double().reshape(-1), sum_dh_xhat.double().reshape(-1)])
    if dist.is_initialized():
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    F = sum_dh.numel()
    return buf[:F], buf[F:]


# ---- sync-BatchNorm wired into the engine (SURVEY.md 8e "BN modes (ii)") -----------------------------------------------------
# The C entry points hand the per-channel partial sums of every BatchNorm (graph-conv layers and head) to a callback between
# the kernel that reduces them and the one that finalizes the statistics -- forward and backward, eager and while a HIP graph is
# being captured (the collective then becomes a node of that graph).  The callback below sums them across the ranks with
# torch.distributed; with it the model equals a reference run at the GLOBAL batch (pad all shards to the same N: padding rows
# enter the per-view BatchNorm through the row count).
_stats_buffers = []           # weak references to the uint8 device tensors the C side carves its scratch from


def register_stats_buffer(t):
    """Tell the hook about a device tensor the C side may hand pieces of to it (engine scratch blocks)."""
    import weakref
    _stats_buffers[:] = [r for r in _stats_buffers if r() is not None]
    if not any(r() is t for r in _stats_buffers):
        _stats_buffers.append(weakref.ref(t))


class StatsAllReducer:
    """ctypes callback `int hook(double* buf, int n, void* stream, void* user)` (include/eagcn_hip.h eagcn_allreduce_fn):
    in-place cross-rank SUM of n doubles that live inside a registered scratch tensor, on the current stream."""

    def __init__(self, group=None):
        from . import _lib as L
        self.group = group
        self.calls = 0
        self.error = None
        self.cfn = L.ALLREDUCE_FN(self._call)          # keep the object alive as long as any C struct points at it

    def world(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def _call(self, ptr, n, stream, user):
        try:
            self.calls += 1
            if not dist.is_initialized():
                return 0
            if dist.get_world_size(self.group) == 1 and os.environ.get('EAGCN_FORCE_DIST', '0') != '1':
                return 0
            view = None
            for r in _stats_buffers:
                t = r()
                if t is None:
                    continue
                base = t.data_ptr()
                if base <= ptr and ptr + 8 * n <= base + t.numel():
                    off = ptr - base
                    view = t[off:off + 8 * n].view(torch.float64)
                    break
            if view is None:
                raise RuntimeError('sync-BatchNorm hook: buffer %#x is not inside a registered scratch tensor' % ptr)
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
            return 0
        except Exception as e:                          # never let an exception cross the C boundary
            self.error = e
            return 1


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

    def active(self):
        """True when there is something to reduce over (more than one rank, or EAGCN_FORCE_DIST=1 for single-GPU tests)."""
        if not dist.is_initialized():
            return False
        return dist.get_world_size(self.group) > 1 or os.environ.get('EAGCN_FORCE_DIST', '0') == '1'

    def agree(self, ok):
        """True iff `ok` holds on EVERY rank (one MIN-reduction of a flag; host sync).  Used once per captured step graph: a
        rank whose capture of the in-graph collective failed must not leave the others replaying collectives it never issues."""
        if not dist.is_initialized():
            return bool(ok)
        dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(self.group) == 'nccl' else torch.device('cpu')
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    def start(self, t):
        """Asynchronous in-place average of a contiguous device tensor (a bucket of the flat gradient buffer): returns the
        work handle; ``.wait()`` orders the current stream behind it.  Capturable: inside a HIP-graph capture the collective
        becomes a branch of the graph that runs beside whatever is issued before the wait."""
        if t.is_cuda:
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        world = dist.get_world_size(self.group)

        class _Scaled:
            def wait(self_inner):
                w.wait()
                t.div_(world)
        return _Scaled()

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
