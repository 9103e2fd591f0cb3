"""HIP-graph execution of the models that run layer by layer: the GAT baseline (reference models.py:69-73, layers.py:99-203)
and the Diff_Pooling read-out (molfp_mode='pool', models.py:104-106, layers.py:492-506).

These models have no model-level C plan (eagcn_model_forward / _backward know the edge-attention layers and the sum / ave
read-out); their step is the sequence of layer-level entry points `EAGCN._forward_composed_index` issues -- the layers, the
read-out, the head (eagcn_head_forward / _backward) -- with autograd in between.  Every kernel of that sequence takes its row counts from device memory and its grid from static
capacities, so for a fixed (model, B, N) the whole step -- forward, fused loss, autograd backward -- is captured ONCE into a
HIP graph over static buffers and replayed:

  * the batch index is built eagerly into a static, capacity-sized index (what reads the caller's tensors stays eager);
  * atom features, sizes, labels, class weights, the loss scale and the dropout seeds live in static device buffers that are
    refilled per batch; the layer kernels read their dropout seed from device memory (eagcn_layer_params.seed_dev,
    eagcn_gat_params.seed_dev), so every replay draws fresh masks;
  * before the sequence is recorded it runs once eagerly with every side effect undone (ComposedRunner._warm_up); every batch,
    the first one included, is then a replay of the graph;
  * the parameter gradients the captured backward produces live in the graph's memory pool and are attached to ``p.grad``
    (or added to what is there) after every replay, as ``loss.backward()`` would have done.

This is the completeness path of the baselines, not the throughput path: no side stream, no double
buffering; the edge-attention models with sum / ave read-out use graph.GraphRunner.
"""
import ctypes as C

import torch

from . import _lib as L
from .graph import StaticIndex
from .losses import _FusedLoss
from .ops import _ptr


class ComposedRunner:
    def __init__(self, model, B, N, channels, device, row_cap=None, edge_cap=None, validate='sync', static_outputs=False):
        self.model, self.device = model, device
        self.key = (B, N, tuple(channels))
        self.validate, self.static_outputs = validate, bool(static_outputs)
        gat = model.structure == 'GAT'
        self.index = StaticIndex(B, N, channels, device, row_cap if row_cap else B * N, edge_cap)
        self.index.bond_lists = gat or self.index.bond_lists
        self.index.c.build_lists = 1 if self.index.bond_lists else 0
        f32 = dict(dtype=torch.float32, device=device)
        nclass = int(model.den3.weight.shape[1])
        self.afm = torch.zeros((B, N, model.n_afeat), **f32)
        self.size = torch.ones(B, dtype=torch.int64, device=device)
        self.labels = torch.zeros((B, nclass), **f32)
        self.weight = torch.ones((nclass, 2), **f32)
        self.scale = torch.ones((), **f32)
        self.scale_is_one = True
        self.seeds = torch.zeros(8, dtype=torch.int64, device=device)
        self.seed_views = [self.seeds[i:i + 1] for i in range(5)]          # four layers + the head's dropout
        self.seeds_host = torch.zeros(8, dtype=torch.int64).pin_memory()
        self.meta_host = torch.zeros(L.META_WORDS, dtype=torch.int32).pin_memory()
        self.pending = None            # event after the last batch's index build + uploads (its meta words are valid behind it)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.train_graphs = {}         # loss kind -> recorded step
        self.eval_graph = None
        self.replays = 0

    # ---- what reads the caller's tensors (eager) ---------------------------------------------------------------------------
    def _raise_if_bad(self, when):
        meta = self.meta_host.tolist()
        if meta[L.META_BAD_ADJ]:
            raise L.EagcnHipError('%s batch held %d adjacency entries outside {0,1}' % (when, meta[L.META_BAD_ADJ]))
        if meta[L.META_BAD_REL]:
            raise L.EagcnHipError('%s batch held %d bonds whose relation channels are not one-hot' % (when, meta[L.META_BAD_REL]))
        if meta[L.META_OVERFLOW]:
            raise L.EagcnHipError('%s batch packed %d rows, more than row_cap=%d' % (when, meta[L.META_OVERFLOW], self.index.T))
        if meta[L.META_EDGE_OVERFLOW]:
            raise L.EagcnHipError('%s batch held %d directed bonds, more than edge_cap=%d' % (when, meta[L.META_EDGE_OVERFLOW], self.index.E))

    def _load(self, adj, rels, afm, size, labels, training):
        lib = L.load()
        if lib.eagcn_gemm_sk_failed():
            raise L.EagcnHipError('a stream-K GEMM hand-off timed out in an earlier step; eagcn_gemm_sk_reset_failed() clears the flag')
        if self.pending is not None:            # the pinned staging buffers are free again; deferred validation of the last batch
            self.pending.synchronize()
            self.pending = None
            if self.validate != 'sync':
                self._raise_if_bad('the previous')
        idx = self.index
        B, N = self.key[0], self.key[1]
        if adj.shape != (B, N, N) or afm.shape != self.afm.shape or len(rels) != idx.K:
            raise L.EagcnHipError('batch tensors adjs %s afms %s (%d relation tensors) do not match the captured shape [%d,%d]'
                                  % (tuple(adj.shape), tuple(afm.shape), len(rels), B, N))
        stream = torch.cuda.current_stream(self.device)
        st = C.c_void_p(stream.cuda_stream)
        idx.c.n_logical = 0
        rel_ptrs = (C.c_void_p * idx.K)(*[r.data_ptr() for r in rels])
        L.check(lib.eagcn_index_build(_ptr(adj), rel_ptrs, idx.ref(), C.c_void_p(self.meta_host.data_ptr()), st),
                'eagcn_index_build')
        L.check(lib.eagcn_index_rows(idx.ref(), st), 'eagcn_index_rows')
        self._keep = (adj, rels)               # (read by the kernels queued above)
        self.afm.copy_(afm, non_blocking=True)
        if size is not None:
            self.size.copy_(size.reshape(-1), non_blocking=True)
        if labels is not None:
            self.labels.copy_(labels.reshape(self.labels.shape), non_blocking=True)
        if training:
            base = int(torch.randint(0, 2 ** 62, (1,)).item())          # host generator: no device synchronisation
            sn = self.seeds_host.numpy()
            for l in range(4):
                sn[l] = (base + 7919 * (l + 1)) & (2 ** 63 - 1)
            sn[4] = (base + 0x51ED27) & (2 ** 63 - 1)
            self.last_seeds = [int(v) for v in sn[:5]]
            self.seeds.copy_(self.seeds_host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(stream)
        self.pending = ev
        if self.validate == 'sync':
            ev.synchronize()
            self.pending = None
            self._raise_if_bad('the')

    # ---- the recorded sequence -------------------------------------------------------------------------------------------------
    def _body(self, kind):
        out, atom, grep = self.model._forward_composed_index(self.index, self.afm, self.size, seeds=self.seed_views)
        if kind is None:
            return None, out, atom, grep
        if kind == 'bce':
            loss = _FusedLoss.apply('bce', out, self.labels, self.weight, self.scale)
        else:
            loss = _FusedLoss.apply('mse', out, self.labels, None, self.scale)
        return loss, out, atom, grep

    def _outputs(self, loss, out, atom_args, grep):
        from .models import LazyAtomRep
        if not self.static_outputs:            # graph_outputs='copy': the caller may keep them across steps
            out, grep = out.clone(), grep.clone()
            loss = loss.clone() if loss is not None else None
        # (the atom representations stay a view of the step's static activations, materialised on first use)
        return loss, (out, LazyAtomRep(*atom_args), grep)

    def train_step(self, adj, rels, afm, size, labels, kind, bce_weight, scale):
        self._load(adj, rels, afm, size, labels, training=True)
        if kind == 'bce':
            self.weight.copy_(bce_weight.reshape(self.weight.shape), non_blocking=True)
        if scale is not None:
            self.scale.copy_(scale, non_blocking=True)
            self.scale_is_one = False
        elif not self.scale_is_one:
            self.scale.fill_(1.0)
            self.scale_is_one = True
        ent = self.train_graphs.get(kind)
        if ent is None:
            ent = self.train_graphs[kind] = self._record(kind)
        keep = []
        for p, g in zip(self.params, ent['grads']):        # ``loss.backward()`` ACCUMULATES into an existing .grad
            keep.append(g.clone() if (g is not None and p.grad is g) else None)
        ent['graph'].replay()
        self.replays += 1
        for p, g, k in zip(self.params, ent['grads'], keep):
            if g is None:
                continue
            if p.grad is None:
                p.grad = g
            elif p.grad is g:
                g.add_(k)
            else:
                p.grad.add_(g)
        return self._outputs(ent['loss'], ent['out'], ent['atom'], ent['grep'])

    def _warm_up(self, kind):
        """One eager pass of the sequence before it is recorded (library handles, workspaces, cached layer descriptions), with
        every side effect undone -- BatchNorm statistics and counters are put back, the gradients it produced are dropped -- and,
        above all, with NOTHING of its autograd graph left alive: a parameter's AccumulateGrad node survives as long as any graph
        references it and remembers the stream it was created on; recorded through such a node the captured backward hops to
        the default stream and the capture dies (segmentation fault inside hipStreamEndCapture, ROCm 7.0)."""
        bufs = list(self.model.buffers())
        snap = [b.clone() for b in bufs]
        if kind is None:
            with torch.no_grad():
                self._body(None)
        else:
            self._body(kind)[0].backward()
            for p in self.params:
                p.grad = None
        with torch.no_grad():
            for b, s in zip(bufs, snap):
                b.copy_(s)

    def _record(self, kind):
        saved = [p.grad for p in self.params]
        for p in self.params:
            p.grad = None
        self._warm_up(kind)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            loss, out, atom, grep = self._body(kind)
            loss.backward()
            loss, out, grep = loss.detach(), out.detach(), grep.detach()
        grads = [p.grad for p in self.params]
        for p, s in zip(self.params, saved):
            p.grad = s
        return dict(graph=g, loss=loss, out=out, atom=atom._args, grep=grep, grads=grads)

    def eval_forward(self, adj, rels, afm, size):
        self._load(adj, rels, afm, size, None, training=False)
        with torch.no_grad():
            if self.eval_graph is None:
                self._warm_up(None)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    _, out, atom, grep = self._body(None)
                self.eval_graph = dict(graph=g, out=out, atom=atom._args, grep=grep)
            ent = self.eval_graph
            ent['graph'].replay()
            self.replays += 1
            _, res = self._outputs(None, ent['out'], ent['atom'], ent['grep'])
            return res

    def stale(self):
        return False

    def release(self):
        self.train_graphs, self.eval_graph = {}, None
