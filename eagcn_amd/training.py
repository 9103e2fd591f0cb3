"""The caller of the hot path, restated for the HIP engine: the inner training loop of train.py:310-334 (zero_grad ->
forward -> MSE or weighted masked BCE-with-logits -> backward -> Adam with weight decay), the periodic evaluation of
train.py:130-211 (eval-mode forward over a loader, sigmoid, per-task ROC-AUC over the labelled entries, or RMSE) and the
class weights of utils.py:681-700.

Everything per-batch stays on the device: the fused loss kernels (csrc/loss.hip) write d loss / d logits straight into the
buffer the captured backward reads, evaluation batches are appended to device-resident score / target / valid buffers by one
small kernel (eagcn_eval_append) and the metrics are computed once per evaluation over those buffers -- the reference moves
every batch to the host and walks B x T Python lists (train.py:147-170)."""
import ctypes as C
import math

import torch

from . import _lib as L
from .losses import fused_classification_loss, fused_regression_loss


class EvalBuffers:
    """Device-resident scores / targets / validity of an evaluation pass ([cap, T]); grows geometrically."""

    def __init__(self, n_tasks, classification, device, cap=4096):
        self.T, self.classification, self.device = int(n_tasks), bool(classification), device
        self.rows = 0
        self._alloc(cap)

    def _alloc(self, cap):
        self.cap = int(cap)
        self.scores = torch.empty((self.cap, self.T), dtype=torch.float32, device=self.device)
        self.targets = torch.empty((self.cap, self.T), dtype=torch.float32, device=self.device)
        self.valid = torch.empty((self.cap, self.T), dtype=torch.uint8, device=self.device)

    def append(self, logits, labels):
        B = logits.shape[0]
        if self.rows + B > self.cap:
            old = (self.scores[:self.rows], self.targets[:self.rows], self.valid[:self.rows])
            self._alloc(max(2 * self.cap, self.rows + B))
            self.scores[:self.rows], self.targets[:self.rows], self.valid[:self.rows] = old
        x = logits.detach().to(torch.float32).contiguous().view(B, self.T)
        y = labels.to(device=x.device, dtype=torch.float32).contiguous().view(B, self.T)
        if not x.is_cuda:
            raise L.EagcnHipError('EvalBuffers.append needs device logits (no CPU path)')
        L.check(L.load().eagcn_eval_append(x.data_ptr(), y.data_ptr(), B, self.T, int(self.classification),
                                           self.scores.data_ptr(), self.targets.data_ptr(), self.valid.data_ptr(),
                                           self.rows, C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'eagcn_eval_append')
        self.rows += B

    def views(self):
        return self.scores[:self.rows], self.targets[:self.rows], self.valid[:self.rows].bool()


def auc_per_task(scores, targets, valid):
    """ROC-AUC of every task over its labelled entries (train.py:156-186: sklearn roc_curve + auc on the entries whose
    label is not -1) as the Mann-Whitney statistic with average ranks for ties -- the same number, computed with tensor ops
    on whatever device the buffers live on.  Returns (list of per-task AUCs with nan for a task without both classes,
    mean over the non-nan tasks)."""
    aucs = []
    for j in range(scores.shape[1]):
        m = valid[:, j]
        s, y = scores[:, j][m].double(), targets[:, j][m]
        pos, neg = (y == 1).sum().item(), (y == 0).sum().item()
        if pos == 0 or neg == 0:
            aucs.append(float('nan'))
            continue
        order = torch.argsort(s)
        ss = s[order]
        ranks = torch.arange(1, ss.numel() + 1, dtype=torch.float64, device=s.device)
        # average rank inside groups of equal scores
        _, inv, counts = torch.unique_consecutive(ss, return_inverse=True, return_counts=True)
        sums = torch.zeros(counts.numel(), dtype=torch.float64, device=s.device).scatter_add_(0, inv, ranks)
        avg = (sums / counts.double())[inv]
        r_pos = avg[(y[order] == 1)].sum().item()
        aucs.append((r_pos - pos * (pos + 1) / 2.0) / (pos * neg))
    good = [a for a in aucs if not math.isnan(a)]
    return aucs, (sum(good) / len(good) if good else float('nan'))


def rmse(scores, targets):
    """train.py:188-211: sqrt of the mean squared error over all outputs."""
    d = scores.double().reshape(-1) - targets.double().reshape(-1)
    return math.sqrt(float((d * d).mean()))


def set_weight(labels, n_tasks=None):
    """Class weights of the weighted BCE, exactly utils.py:681-700: per task [5000 / #positives, 5000 / #negatives] over the
    labels of the training set (array-like [n, T]; anything other than 0 / 1 is a missing label).  Returned as a list
    indexed by task, what train.py hands to ``weight_tensor`` (utils.py:653-679) as ``weights[j][0]`` / ``weights[j][1]``.
    A task without a positive label has no entry in the reference's dictionary and its training loop dies with a KeyError
    as soon as a label of that task shows up (a task with positives but no negatives already in set_weight itself); the
    same error is raised here, at once, naming the task."""
    y = torch.as_tensor(labels)
    if y.dim() != 2:
        raise ValueError('set_weight: labels must be [n, tasks], got %s' % (tuple(y.shape),))
    T = y.shape[1] if n_tasks is None else int(n_tasks)
    out = []
    for j in range(T):
        pos, neg = int((y[:, j] == 1).sum()), int((y[:, j] == 0).sum())
        if pos == 0 or neg == 0:
            raise KeyError('set_weight: task %d has %d positive and %d negative labels (utils.py:697-699 has no weight '
                           'for it)' % (j, pos, neg))
        out.append([5000 / pos, 5000 / neg])
    return out


def train_step(model, optimizer, batch, labels, task, bce_weight=None, dp_global_norm=False, fused=None, bonds=None,
               reducer=None):
    """One iteration of train.py:310-334.  `batch` = (adjs, afms, TypeAtt, OrderAtt, AromAtt, ConjAtt, RingAtt, size) device
    tensors -- or (afms, size) together with `bonds` (a CompactBonds: eagcn_amd.collate.collate_compact) for a compact batch;
    returns the loss tensor (device; no host sync).  With a graph-mode model the three phases run as ONE captured graph
    (EAGCN.fused_step) unless fused=False.

    Data parallel: `reducer` (an eagcn_amd.parallel.GradientAllReducer over the model's parameters) averages the gradients
    across ranks between backward and ``optimizer.step()``.  ``dp_global_norm=True`` rescales this rank's loss by
    world * n_r / sum n_r (parallel.dp_loss_scale), which is only meaningful when the gradients are averaged afterwards: with
    torch.distributed initialised it therefore REQUIRES a reducer (replicas that never synchronise and step on mis-scaled
    gradients would be the silent alternative)."""
    if dp_global_norm and reducer is None:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            raise ValueError('train_step(dp_global_norm=True) under torch.distributed needs reducer=GradientAllReducer(...): '
                             'the global loss normalisation assumes the gradients are averaged across ranks')
    optimizer.zero_grad(set_to_none=True)
    if fused is None:
        fused = bool(getattr(model, 'graph', False)) and model.training and hasattr(model, 'fused_step')
        # the captured step of the GAT / pool models (graph_composed) takes no reducer and no 'dp' scale: those go through the
        # unfused path below, which handles both
        composed = getattr(model, 'structure', None) == 'GAT' or getattr(model, 'molfp_mode', None) == 'pool'
        if fused and composed and (reducer is not None or dp_global_norm):
            fused = False
    if fused:
        # the global BCE normalisation ('dp': a 1-element collective issued with the batch's preparatory work) and the gradient
        # average (captured into the step graph, upper bucket overlapped with the first layer's backward) are part of the step
        from .optim import FlatAdam
        composed = getattr(model, 'structure', None) == 'GAT' or getattr(model, 'molfp_mode', None) == 'pool'
        in_graph = isinstance(optimizer, FlatAdam) and not composed      # the update is the last launch of the step graph
        loss, _ = model.fused_step(batch, labels, task, bce_weight, 'dp' if dp_global_norm else None, bonds=bonds, reducer=reducer,
                                   **({'optimizer': optimizer} if in_graph else {}))
        if not in_graph:
            optimizer.step()
        return loss
    else:
        out, _, _ = model(*batch) if bonds is None else model.forward_compact(bonds, *batch)
        if task == 'reg':
            loss = fused_regression_loss(out, labels)
        else:
            loss = fused_classification_loss(out, labels, bce_weight, dp_global_norm=dp_global_norm,
                                             group=reducer.group if reducer is not None else None)
        loss.backward()
    if reducer is not None:
        reducer()
    optimizer.step()
    return loss


@torch.no_grad()
def evaluate(model, batches, task, n_tasks):
    """train.py:130-211: eval-mode forward over `batches` (iterable of (batch tuple, labels); a batch tuple that starts with a
    CompactBonds is (bonds, afms, size)); classification returns (per-task AUCs, mean AUC), regression the RMSE.  The model
    is put back into training mode afterwards, as the reference does (train.py:171, 210)."""
    was_training = model.training
    model.eval()
    buf = None
    for batch, labels in batches:
        compact = hasattr(batch[0], 'bond_mol')
        out, _, _ = model.forward_compact(*batch) if compact else model(*batch)
        if buf is None:
            buf = EvalBuffers(n_tasks, task != 'reg', out.device)
        buf.append(out, labels)
    model.train(was_training)
    scores, targets, valid = buf.views()
    return auc_per_task(scores, targets, valid) if task != 'reg' else rmse(scores, targets)
