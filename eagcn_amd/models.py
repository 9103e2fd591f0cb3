"""EAGCN model (reference eagcn_pytorch/models.py:14-121) on the HIP hot path.

Same constructor / forward signature and state_dict keys as the reference's ``EAGCN``; the class
names used by BASELINE.json (``Concate_GCN``, ``Weighted_GCN``) are provided as thin subclasses
that fix ``structure``.  Keyword-only extensions (defaults = reference behaviour):

  n_layers      number of graph-conv layers, 1..4 (reference hard-codes 4, models.py:50-61);
                BASELINE.json's configs use 2 and 3.  Layers 3/4 use twice the layer-2 widths.
  rel_channels  channels of every relation tensor (default [n_bfeat,4,2,2,2]); with ``widths1`` /
                ``widths2`` lists this allows K != 5 views.
  grad_mode     'autograd' (default): parameter gradients are returned through autograd as usual;
                'direct': the backward call writes them into ONE flat buffer and sets ``p.grad`` to
                views of it (accumulating into an existing ``.grad``), which skips ~70 AccumulateGrad
                nodes per step.  Same values; tensor hooks on parameters do not fire in this mode.
  graph         False (default) | True: replay the training step as two captured HIP graphs (forward,
                backward) instead of ~60 eager launches; see eagcn_amd/graph.py.  Needs a fixed (B, N)
                per captured pair (a new pair is captured for every new shape), delivers gradients
                as in grad_mode='direct', keeps ONE training forward in flight, and the lazy atom
                representations are valid until the next forward.  ``row_cap`` bounds the packed rows
                the static buffers are sized for (default B*N); ``n_bucket`` (e.g. 16) rounds the padded size N up
                to a multiple so that the batches of a real loader -- the reference pads each batch to ITS maximum,
                utils.py:583 -- share one runner per bucket instead of one per distinct N (the kernels still see the
                batch's own N: BatchNorm row counts and filler weights come from a device word); ``edge_cap`` the directed bonds of a batch
                (default 8 per row of row_cap: molecular graphs hold 2-2.5).  Every distinct (B, N, training) shape owns a
                runner (two index slots, two saved-activation blocks, scratch: see GraphRunner.nbytes());
                at most ``max_runners`` (default 8) are kept, least recently used evicted first.
  graph_outputs 'copy' (default): graph mode returns fresh tensors like the reference does; 'static': it
                returns views of the runner's static output buffers, which the NEXT forward of the same shape
                overwrites (saves two small copies per step; for loops that consume the outputs at once).
  validate      'sync' (default): graph mode waits for the batch-index kernels of the batch it was given and
                raises on a non-binary adjacency / non-one-hot relation tensor / row_cap overflow BEFORE
                the forward runs; 'deferred': no host wait at all, the same errors are raised when a later
                batch is submitted (one or two steps late) -- for input pipelines that are known to be valid.
  overlap_index False (default) | True: declare that the batch tensors are already resident in HBM when
                forward is called (prefetched batches); the batch index then runs on a side stream
                without waiting for the previous step's queued work (see ops.BatchIndex).
  atom_rep      'lazy' (default) | 'eager' | 'none': the reference copies the last layer's atom
                representations to the host in EVERY forward (``x2.data.cpu()``, models.py:102), a
                device->host copy plus a sync per step; 'lazy' returns an object that performs the
                copy on first use, 'eager' reproduces the reference exactly.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .layers import GAT, Dense, Diff_Pooling, GraphConv_Layer, Vanilla_GCN


class LazyAtomRep:
    """Deferred ``x2.data.cpu()`` (models.py:102): materialises the padded [B,N,F] host tensor on
    first use.  Supports what train.py:213-266 does with it (.view / indexing / .numpy / .shape)."""

    def __init__(self, index, layout, packed, pad_row, n_out=None, materialize=None):
        self._args = (index, layout, packed.detach(), None if pad_row is None else pad_row.detach())
        self._cpu = None
        self._n_out = n_out            # N of the caller's batch when the index was built for a larger (bucketed) N
        self._materialize = materialize    # builds `packed` on the device first (the forward skipped it: fused read-out)

    def _ensure(self):
        if self._materialize is not None:
            fn, self._materialize = self._materialize, None
            fn()

    @property
    def packed(self):
        """(packed [T, ld] device tensor, pad_row, layout) without the padded host copy."""
        self._ensure()
        index, layout, packed, pad_row = self._args
        return packed, pad_row, layout

    def cpu(self):
        if self._cpu is None:
            self._ensure()
            index, layout, packed, pad_row = self._args
            with torch.no_grad():
                dense = ops.unpack_rows(index, layout, packed, pad_row)
                self._cpu = (dense if self._n_out is None else dense[:, :self._n_out]).cpu()
            self._args = None
        return self._cpu

    def __getattr__(self, name):
        return getattr(self.cpu(), name)

    def __getitem__(self, i):
        return self.cpu()[i]


class EAGCN(nn.Module):
    def __init__(self, n_bfeat, n_afeat, n_sgc1_1=None, n_sgc1_2=None, n_sgc1_3=None, n_sgc1_4=None,
                 n_sgc1_5=None, n_sgc2_1=None, n_sgc2_2=None, n_sgc2_3=None, n_sgc2_4=None, n_sgc2_5=None,
                 n_den1=128, n_den2=64, nclass=1, dropout=0.0, structure='Concate', molfp_mode='sum',
                 pool_num=5, *, n_layers=4, widths1=None, widths2=None, rel_channels=None, atom_rep='lazy',
                 grad_mode='autograd', overlap_index=False, graph=False, row_cap=None, graph_outputs='copy',
                 validate='sync', max_runners=8, edge_cap=None, n_bucket=0, relations='onehot', sync_bn=False):
        super().__init__()
        self.n_bucket = int(n_bucket)
        self.sync_bn = bool(sync_bn)
        if relations not in ('onehot', 'general'):
            raise ValueError("relations must be 'onehot' (neural_fp.py:111-120) or 'general' (any channel values, layers.py:82)")
        self.relations = relations
        if widths1 is None:
            widths1 = [n_sgc1_1, n_sgc1_2, n_sgc1_3, n_sgc1_4, n_sgc1_5]
        if widths2 is None:
            widths2 = [n_sgc2_1, n_sgc2_2, n_sgc2_3, n_sgc2_4, n_sgc2_5]
        widths1, widths2 = [int(w) for w in widths1], [int(w) for w in widths2]
        K = len(widths1)
        if len(widths2) != K:
            raise ValueError('widths1 and widths2 need one entry per view')
        if rel_channels is None:
            rel_channels = [n_bfeat, 4, 2, 2, 2][:K]
        if structure not in ('Concate', 'Weighted_sum', 'GCN', 'GAT'):
            raise ValueError("structure must be 'Concate', 'Weighted_sum' or one of the baselines 'GCN' / 'GAT' (models.py:50-73)")
        if molfp_mode not in ('sum', 'ave', 'pool'):
            raise ValueError("molfp_mode must be 'sum', 'ave' or 'pool' (models.py:104-111)")
        if molfp_mode == 'pool':
            if structure in ('Concate', 'Weighted_sum') and n_layers != 4:
                raise ValueError("molfp_mode='pool' reads the attention matrix of layer 4 (last=True, layers.py:319-324)")
        self.pool_num = int(pool_num)
        if sync_bn and (structure == 'GAT' or molfp_mode == 'pool'):
            raise ValueError("sync_bn is wired into the model engine (structure Concate / Weighted_sum / GCN, molfp_mode sum / ave)")
        if not 1 <= n_layers <= 4:
            raise ValueError('n_layers must be 1..4')
        if structure == 'GAT':                                            # models.py:69-73: four GAT layers
            self.ngc1, self.ngc2 = sum(widths1), sum(widths2)
            gplan = [(n_afeat, self.ngc1), (self.ngc1, self.ngc2), (self.ngc2, self.ngc2), (self.ngc2, 2 * self.ngc2)][:n_layers]
            for i, (fin, fout) in enumerate(gplan):
                setattr(self, 'layer%d' % (i + 1), GAT(fin, fout, dropout))
            self._finish_init(n_layers, n_afeat, 1, structure, molfp_mode, dropout, atom_rep, grad_mode, overlap_index, graph,
                              row_cap, edge_cap, graph_outputs, validate, max_runners, gplan[-1][1], n_den1, n_den2, nclass)
            return
        if structure == 'GCN':                                            # models.py:63-67: four Vanilla_GCN layers
            self.ngc1, self.ngc2 = sum(widths1), sum(widths2)
            gplan = [(n_afeat, self.ngc1), (self.ngc1, self.ngc2), (self.ngc2, self.ngc2), (self.ngc2, 2 * self.ngc2)][:n_layers]
            for i, (fin, fout) in enumerate(gplan):
                setattr(self, 'layer%d' % (i + 1), Vanilla_GCN(fin, fout, dropout, bond_channels=rel_channels[0]))
            self._finish_init(n_layers, n_afeat, 1, structure, molfp_mode, dropout, atom_rep, grad_mode, overlap_index, graph,
                              row_cap, edge_cap, graph_outputs, validate, max_runners, gplan[-1][1], n_den1, n_den2, nclass)
            return
        if structure == 'Weighted_sum':                                   # models.py:33-47
            widths1 = [sum(widths1)] * K
            widths2 = [sum(widths2)] * K
            self.ngc1, self.ngc2 = widths1[0], widths2[0]
        else:
            self.ngc1, self.ngc2 = sum(widths1), sum(widths2)
        w3 = [2 * w for w in widths2]
        plan = [(n_afeat, widths1, self.ngc1), (self.ngc1, widths2, self.ngc2),
                (self.ngc2, w3, 2 * self.ngc2), (2 * self.ngc2, w3, 2 * self.ngc2)][:n_layers]
        for i, (fin, ws, _) in enumerate(plan):
            setattr(self, 'layer%d' % (i + 1),
                    GraphConv_Layer(fin, n_bfeat, None, dropout=dropout, structure=structure, last=(i == 3),
                                    widths=ws, rel_channels=rel_channels))
        self._finish_init(n_layers, n_afeat, K, structure, molfp_mode, dropout, atom_rep, grad_mode, overlap_index, graph,
                          row_cap, edge_cap, graph_outputs, validate, max_runners, plan[-1][2], n_den1, n_den2, nclass)

    def _finish_init(self, n_layers, n_afeat, K, structure, molfp_mode, dropout, atom_rep, grad_mode, overlap_index, graph,
                     row_cap, edge_cap, graph_outputs, validate, max_runners, f_last, n_den1, n_den2, nclass):
        self.n_layers, self.n_afeat, self.K = n_layers, n_afeat, K
        self.structure, self.molfp_mode, self.dropout = structure, molfp_mode, dropout
        self.atom_rep = atom_rep
        if grad_mode not in ('autograd', 'direct'):
            raise ValueError("grad_mode must be 'autograd' or 'direct'")
        self.grad_mode = grad_mode
        self.overlap_index = bool(overlap_index)
        self.graph, self.row_cap, self.edge_cap = bool(graph), row_cap, edge_cap
        if graph_outputs not in ('copy', 'static'):
            raise ValueError("graph_outputs must be 'copy' or 'static'")
        if validate not in ('sync', 'deferred'):
            raise ValueError("validate must be 'sync' or 'deferred'")
        self.graph_outputs, self.validate, self.max_runners = graph_outputs, validate, int(max_runners)
        self._runners = {}
        self.den1 = Dense(f_last, n_den1)
        self.den2 = Dense(n_den1, n_den2)
        self.den3 = Dense(n_den2, nclass)
        self.Graph_BN = nn.BatchNorm1d(f_last)
        self.bn_den1 = nn.BatchNorm1d(n_den1)
        self.bn_den2 = nn.BatchNorm1d(n_den2)
        if molfp_mode == 'pool':                                          # models.py:90-92 (pool3 is built, never called)
            self.pool1 = Diff_Pooling(f_last, f_last, self.pool_num)
            self.pool3 = Diff_Pooling(f_last, f_last, 1)
        self._plan = None

    def graph_layers(self):
        return [getattr(self, 'layer%d' % (i + 1)) for i in range(self.n_layers)]

    def forward_layers(self, index, afms, seeds=None):
        """Packed activations after every graph-conv layer: list of (x, pad_row, layout).  `seeds`: one dropout seed per layer
        (ints, or 1-element int64 device tensors -- graph mode: the kernels read them from device memory); default: drawn."""
        layout = ops.ColLayout.single(self.n_afeat)
        x = ops.pack_rows(index, layout, afms)
        outs = []
        for l, layer in enumerate(self.graph_layers()):
            x, pad_row, layout = layer.forward_packed(index, x, layout, None if seeds is None else seeds[l])
            outs.append((x, pad_row, layout))
        return outs

    def plan(self):
        if self.molfp_mode == 'pool':
            raise ops.L.EagcnHipError("molfp_mode='pool' has no model-level plan: it runs layer by layer (forward_composed)")
        if self._plan is None:
            head = {n: getattr(self, n) for n in ('den1', 'den2', 'den3', 'Graph_BN', 'bn_den1', 'bn_den2')}
            stats = None
            if getattr(self, 'sync_bn', False):       # every BatchNorm over the GLOBAL batch (SURVEY.md 8e "BN modes (ii)")
                from .parallel import StatsAllReducer
                stats = StatsAllReducer()
            # atom_rep 'lazy' / 'none': the top layer's output matrix is built only if somebody reads the atom representations
            fuse = self.atom_rep in ('lazy', 'none') and self.structure == 'Concate'
            self._plan = ops.ModelPlan(self.graph_layers(), head, self.n_afeat, self.molfp_mode, self.dropout, stats, fuse)
        return self._plan

    # The plan and the graph runners hold ctypes structs with device pointers, captured HIP graphs and static
    # buffers: none of that can (or should) travel with a pickled / deep-copied module -- the reference pickles
    # the whole model at every best-validation checkpoint (train.py:440, torch.save(model, ...)).  They are
    # dropped here and rebuilt lazily by the first forward of the copy.
    def __getstate__(self):
        if self._plan is not None:
            self._plan.flush_nbt()
        state = super().__getstate__() if hasattr(nn.Module, '__getstate__') else self.__dict__.copy()
        state = dict(state)
        state['_plan'] = None
        state['_runners'] = {}
        state.pop('_bce_weight_cache', None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._plan = None
        self._runners = {}
        for name, default in (('graph_outputs', 'copy'), ('validate', 'sync'), ('max_runners', 8), ('edge_cap', None), ('n_bucket', 0),
                              ('relations', 'onehot'), ('pool_num', 5), ('sync_bn', False)):
            self.__dict__.setdefault(name, default)

    def state_dict(self, *a, **kw):
        if self._plan is not None:
            self._plan.flush_nbt()                      # graph mode counts num_batches_tracked on the host
        return super().state_dict(*a, **kw)

    def load_state_dict(self, *a, **kw):
        if self._plan is not None:
            self._plan.flush_nbt()                      # pending counts belong to the values being replaced
        return super().load_state_dict(*a, **kw)

    def _apply(self, fn, *a, **kw):                     # .cuda() / .to(): parameters are re-created
        if self._plan is not None:
            self._plan.flush_nbt()
        self._plan = None
        self._runners = {}
        return super()._apply(fn, *a, **kw)

    def _graph_runner(self, adjs, afms, rels, size, bonds=None):
        """Validated inputs and the (cached, least-recently-used) GraphRunner of this batch shape."""
        from . import graph as G
        afms = ops._need_cuda_f32(afms, 'afms')
        if bonds is None:
            adjs = ops._need_cuda_f32(adjs, 'adjs')
            rels = [ops._need_cuda_f32(r, 'relation tensor %d' % i) for i, r in enumerate(rels)]
            B, N = adjs.shape[0], adjs.shape[1]
            if adjs.dim() != 3 or adjs.shape[2] != N or afms.shape != (B, N, self.n_afeat) or len(rels) != self.K:
                raise ops.L.EagcnHipError('inconsistent batch tensors: adjs %s afms %s, %d relation tensors'
                                          % (tuple(adjs.shape), tuple(afms.shape), len(rels)))
            for i, r in enumerate(rels):
                if r.dim() != 4 or r.shape[0] != B or r.shape[2] != N or r.shape[3] != N:
                    raise ops.L.EagcnHipError('relation tensor %d must be [B,C,N,N], got %s' % (i, tuple(r.shape)))
            channels = tuple(int(r.shape[1]) for r in rels)
            btuple = None
        else:
            B, N, channels = bonds.B, bonds.N, tuple(bonds.channels)
            if afms.shape != (B, N, self.n_afeat) or len(channels) != self.K:
                raise ops.L.EagcnHipError('inconsistent compact batch: afms %s for B=%d N=%d, %d views'
                                          % (tuple(afms.shape), B, N, len(channels)))
            self._check_channels(channels, bonds.rel_vectors)
            btuple = bonds.checked()
            if bonds.rel_vectors is not None:
                # general relation vectors: the runner's index holds a 255-row code book per view (static buffers, refilled per
                # batch), so one pair of graphs serves batches with different numbers of distinct vectors
                general = tuple(int(v.shape[1]) for v in bonds.rel_vectors)
                channels = tuple(255 for _ in general)
        plan = self.plan()
        if bonds is None or bonds.rel_vectors is None:
            general = None
        n_in = N
        if self.n_bucket > 1:                       # one runner (one pair of captured graphs) per BUCKET of padded sizes
            N = -(-N // self.n_bucket) * self.n_bucket
        key = (B, N, channels, float(self.dropout), self.training, general)
        runner = self._runners.pop(key, None)
        if runner is None or runner.stale():
            while len(self._runners) >= max(1, self.max_runners):       # least recently used first (dict order)
                old = self._runners.pop(next(iter(self._runners)))
                old.release()
            runner = G.GraphRunner(plan, B, N, channels, afms.device, self.dropout, self.row_cap, training=self.training,
                                   static_outputs=(self.graph_outputs == 'static'), validate=self.validate,
                                   edge_cap=self.edge_cap, rel_c=general)
        self._runners[key] = runner                                      # (re-)inserted last = most recently used
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (self.dropout > 0 and self.training) else 0
        if self.molfp_mode == 'ave':
            size = size.to(device=afms.device, dtype=torch.int64)
        runner.n_in = n_in                           # padded size of THIS batch's tensors (<= the runner's capacity N)
        runner.rel_vectors = None if general is None else bonds.rel_vectors
        return runner, adjs, rels, afms, size, seed, btuple

    def _composed_runner(self, adjs, afms, rels, size):
        """Validated dense inputs and the cached graph_composed.ComposedRunner of this batch shape (GAT / pool models)."""
        from .graph_composed import ComposedRunner
        adjs = ops._need_cuda_f32(adjs, 'adjs')
        afms = ops._need_cuda_f32(afms, 'afms')
        if self.structure in ('GCN', 'GAT'):
            rels = rels[:1]
        rels = [ops._need_cuda_f32(r, 'relation tensor %d' % i) for i, r in enumerate(rels)]
        B, N = adjs.shape[0], adjs.shape[1]
        if adjs.dim() != 3 or adjs.shape[2] != N or afms.shape != (B, N, self.n_afeat) or len(rels) != self.K:
            raise ops.L.EagcnHipError('inconsistent batch tensors: adjs %s afms %s, %d relation tensors'
                                      % (tuple(adjs.shape), tuple(afms.shape), len(rels)))
        channels = tuple(int(r.shape[1]) for r in rels)
        if self.structure not in ('GCN', 'GAT'):
            self._check_channels(channels)
        key = ('composed', B, N, channels)
        runner = self._runners.pop(key, None)
        if runner is None:
            while len(self._runners) >= max(1, self.max_runners):
                self._runners.pop(next(iter(self._runners))).release()
            runner = ComposedRunner(self, B, N, channels, afms.device, self.row_cap, self.edge_cap, self.validate,
                                    static_outputs=(self.graph_outputs == 'static'))
        self._runners[key] = runner
        if size is not None and isinstance(size, torch.Tensor):
            size = size.to(device=afms.device, dtype=torch.int64)
        return runner, adjs, rels, afms, size

    def _atom_rep(self, runner):
        if self.atom_rep == 'none':
            return None
        pad = runner.pad_view if self.structure in ('Weighted_sum', 'GCN') else None
        rep = LazyAtomRep(runner.index, self.plan().last_layout, runner.xout_view, pad,
                          runner.n_in if runner.n_in != runner.key[1] else None, runner.materializer())
        return rep.cpu() if self.atom_rep == 'eager' else rep

    def _graph_forward(self, adjs, afms, rels, size, bonds=None):
        from . import graph as G
        runner, adjs, rels, afms, size, seed, btuple = self._graph_runner(adjs, afms, rels, size, bonds)
        out, graph_representation = G.graph_forward(runner, adjs, rels, afms, size, seed, self.overlap_index, btuple)
        return out, self._atom_rep(runner), graph_representation

    def fused_step(self, batch, labels, task, bce_weight=None, scale=None, bonds=None, reducer=None, optimizer=None):
        """forward -> loss -> backward of one training batch as ONE captured graph launch (graph mode only): the inner
        loop of train.py:310-334 without the launch boundaries between the three phases.  `batch` is the reference's
        forward argument tuple (adjs, afms, TypeAtt, ..., size) -- or (afms, size) together with `bonds` for a compact
        batch; task 'reg' = MSE (train.py:321-325), anything else = weighted masked BCE-with-logits (train.py:326-331)
        with `bce_weight` [T,2]; `scale` an optional device scalar multiplied into loss and gradient (data-parallel
        global normalisation, parallel.dp_loss_scale) -- or the string 'dp': the same factor, computed from this batch's
        labels by a 1-element collective issued with the batch's other preparatory work (on the side stream under the
        previous step when overlap_index is on).  `reducer` (parallel.GradientAllReducer): the gradient average over
        the ranks is part of the step -- captured INTO the step graph, the bucket of the upper layers + head starting
        while the first layer's backward still runs; the caller does not call the reducer again.  `optimizer` (an
        eagcn_amd.optim.FlatAdam over this model): the parameter update (train.py:334) is the last launch of the same graph --
        the caller does not call ``optimizer.step()``; gradients must not be accumulated across steps then.  Returns (loss, (out,
        atom_representations, graph_representation)); the parameter gradients are attached to ``p.grad`` as
        ``loss.backward()`` would."""
        if not (self.graph and self.training and torch.is_grad_enabled()):
            raise ops.L.EagcnHipError('fused_step needs graph=True, training mode and grad enabled')
        if self.structure == 'GAT' or self.molfp_mode == 'pool':
            # no model-level plan: the layer-by-layer step (forward_composed + fused loss + autograd backward) is captured as
            # one graph over static buffers (graph_composed.ComposedRunner)
            if bonds is not None or reducer is not None or isinstance(scale, str) or optimizer is not None:
                raise ops.L.EagcnHipError("fused_step of a GAT / pool model takes the dense batch, no reducer / optimizer and a tensor scale")
            adjs, afms, *rels_and_size = batch
            *rels, size = rels_and_size
            runner, adjs, rels, afms, size = self._composed_runner(adjs, afms, rels, size)
            kind = 'mse' if task == 'reg' else 'bce'
            if kind == 'bce':
                if bce_weight is None:
                    raise ops.L.EagcnHipError('fused_step: the classification loss needs bce_weight ([T,2]; training.set_weight)')
                if not isinstance(bce_weight, torch.Tensor):
                    bce_weight = torch.tensor(bce_weight, dtype=torch.float32, device=afms.device)
            return runner.train_step(adjs, rels, afms, size, labels, kind, bce_weight, scale)
        if bonds is None:
            adjs, afms, *rels_and_size = batch
            *rels, size = rels_and_size
            if self.structure == 'GCN':
                rels = rels[:1]
        else:
            adjs, rels, (afms, size) = None, None, batch
            if self.structure == 'GCN':
                bonds = bonds.first_view()
        runner, adjs, rels, afms, size, seed, btuple = self._graph_runner(adjs, afms, rels, size, bonds)
        kind = 'mse' if task == 'reg' else 'bce'
        if kind == 'bce' and not isinstance(bce_weight, torch.Tensor):
            if bce_weight is None:
                raise ops.L.EagcnHipError('fused_step: the classification loss needs bce_weight ([T,2]; training.set_weight)')
            # the list set_weight returns (utils.py:681-700), as the eager losses accept it; converted once per CONTENT (2 T floats:
            # a list object's id can be recycled, and a list can be edited in place)
            key = tuple(tuple(float(v) for v in row) for row in bce_weight)
            if getattr(self, '_bce_weight_cache', (None, None))[0] != key:
                self._bce_weight_cache = (key, torch.tensor(bce_weight, dtype=torch.float32, device=afms.device))
            bce_weight = self._bce_weight_cache[1]
        if optimizer is not None and (getattr(optimizer, 'model', None) is not self or not hasattr(optimizer, 'launch')):
            raise ops.L.EagcnHipError('fused_step(optimizer=...) takes the eagcn_amd.optim.FlatAdam built over this model')
        loss = runner.train_step(adjs, rels, afms, size, seed, labels, kind, bce_weight, scale, self.overlap_index, btuple,
                                 comm=reducer, optimizer=optimizer)
        out, graph_representation = runner.outputs()
        return loss, (out, self._atom_rep(runner), graph_representation)

    def forward(self, adjs, afms, *rels_and_size):
        """Reference signature (models.py:96): (adjs, afms, TypeAtt, OrderAtt, AromAtt, ConjAtt, RingAtt,
        size) -> (x, atom_representations, graph_representation).  The whole forward is one call into
        eagcn_model_forward (layers, read-out and head); backward is one call into eagcn_model_backward."""
        *rels, size = rels_and_size
        if self.relations == 'general' and self.structure in ('Concate', 'Weighted_sum'):
            # relation tensors with arbitrary channel values (layers.py:82): code books from the distinct channel vectors at
            # the bonds, then the compact path (the dense tensors are read once, by the canonicalisation)
            from .collate import bonds_from_dense
            adjs = ops._need_cuda_f32(adjs, 'adjs')
            return self.forward_compact(bonds_from_dense(adjs, rels), afms, size)
        if self.structure == 'GAT' or self.molfp_mode == 'pool':     # layer-level entry points + composed head
            if self.graph and not self.training and not torch.is_grad_enabled():
                runner, adjs, rels, afms, size = self._composed_runner(adjs, afms, rels, size)
                return runner.eval_forward(adjs, rels, afms, size)   # forward-only graph (train.py:130-211 under no_grad)
            return self.forward_composed(adjs, afms, *rels, size)
        if self.structure == 'GCN':
            rels = rels[:1]                                          # Vanilla_GCN only needs the bond positions (= adj)
        if self.graph and (self.training and torch.is_grad_enabled() or not self.training and not torch.is_grad_enabled()):
            return self._graph_forward(adjs, afms, rels, size)       # training step, or eval under no_grad (train.py:130-211)
        self._check_channels([int(r.shape[1]) for r in rels])
        index = ops.BatchIndex(adjs, rels, overlap=self.overlap_index,    # once per batch, shared by all layers
                               structure={'Concate': 0, 'Weighted_sum': 1}.get(self.structure, -1))
        return self._forward_index(index, afms, size)

    def forward_compact(self, bonds, afms, size):
        """The same forward from a COMPACT batch: ``bonds`` is an ``eagcn_amd.synthetic.CompactBonds``
        (directed bond list + per-view bond type), ``afms`` the padded [B,N,n_afeat] atom features.  Results are
        identical to ``forward`` on the dense tensors the reference's collate (utils.py:575-640) would build for
        the same molecules; the adjacency / relation tensors are never materialised."""
        if self.structure in ('GCN', 'GAT'):
            bonds = bonds.first_view()
        if self.structure == 'GAT':
            index = ops.BatchIndex.from_bonds(bonds.B, bonds.N, bonds.channels, *bonds.checked(), bond_lists=True)
            return self._forward_composed_index(index, ops._need_cuda_f32(afms, 'afms'), size)
        if self.molfp_mode == 'pool':
            self._check_channels(bonds.channels, bonds.rel_vectors)
            index = ops.BatchIndex.from_bonds(bonds.B, bonds.N, bonds.channels, *bonds.checked(), rel_vectors=bonds.rel_vectors)
            return self._forward_composed_index(index, ops._need_cuda_f32(afms, 'afms'), size)
        if self.graph and (self.training and torch.is_grad_enabled() or not self.training and not torch.is_grad_enabled()):
            return self._graph_forward(None, afms, None, size, bonds)
        self._check_channels(bonds.channels, bonds.rel_vectors)
        index = ops.BatchIndex.from_bonds(bonds.B, bonds.N, bonds.channels, *bonds.checked(), rel_vectors=bonds.rel_vectors)
        return self._forward_index(index, afms, size)

    def _check_channels(self, channels, rel_vectors=None):
        """The attention weight of view k has rel_channels[k] entries (layers.py:64): a one-hot batch may use that many bond
        types, a general batch (rel_vectors) must bring channel vectors of exactly that length."""
        want = self.graph_layers()[0].rel_channels
        if len(channels) != len(want):
            raise ops.L.EagcnHipError('batch has %d relation views, the model %d' % (len(channels), len(want)))
        for k, (c, w) in enumerate(zip(channels, want)):
            if rel_vectors is not None:
                if int(rel_vectors[k].shape[1]) != int(w):
                    raise ops.L.EagcnHipError('view %d: relation vectors have %d channels, the attention weight %d'
                                              % (k, int(rel_vectors[k].shape[1]), w))
            elif int(c) > int(w):
                raise ops.L.EagcnHipError('view %d: %d bond types but the attention weight has %d channels' % (k, c, w))

    def _forward_index(self, index, afms, size):
        plan = self.plan()
        seed = 0
        if self.training and self.dropout > 0:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        holder = {} if self.atom_rep != 'none' else None
        out, graph_representation = ops.model_forward(plan, index, holder, self.training, seed, self.dropout,
                                                      size, afms, direct=(self.grad_mode == 'direct'))
        if self.training:
            torch._foreach_add_(plan.nbt, 1)
        atom_representations = None
        if holder is not None:
            pad = holder['pad_row'] if self.structure in ('Weighted_sum', 'GCN') else None
            atom_representations = LazyAtomRep(index, plan.last_layout, holder['xout'], pad, None, holder.get('materialize'))
            if self.atom_rep == 'eager':
                atom_representations = atom_representations.cpu()
        return out, atom_representations, graph_representation

    def release_graphs(self):
        """Destroy every captured HIP graph and static buffer of this model NOW (they are rebuilt by the next forward).  Orderly
        shutdown of a data-parallel run: the step graphs hold RCCL kernels and must be gone before the process group."""
        runners, self._runners = self._runners, {}
        for r in runners.values():
            r.release()

    def flat_grad_buffer(self):
        """The single fp32 buffer that holds every hot-path parameter gradient after a backward in
        grad_mode='direct' or graph mode (``p.grad`` are views of it); None otherwise."""
        def first_live(plan):            # (frozen buffers in plan.params never carry a .grad)
            return next((i for i, p in enumerate(plan.params) if p.requires_grad), None)
        if self.graph and self._runners:
            for r in self._runners.values():
                if not hasattr(r, 'flat_acc'):       # graph_composed.ComposedRunner (GAT / pool): gradients live in its graph's pool
                    continue
                i = first_live(r.plan)
                if i is not None and r.plan.params[i].grad is r.acc_views[i]:
                    return r.flat_acc
        if self._plan is not None and self._plan.flat_grad is not None:
            i = first_live(self._plan)
            g0 = self._plan.params[i].grad if i is not None else None
            if g0 is not None and self._plan.flat_grad.data_ptr() <= g0.data_ptr() < \
                    self._plan.flat_grad.data_ptr() + 4 * self._plan.flat_grad.numel():
                return self._plan.flat_grad
        return None

    def forward_composed(self, adjs, afms, *rels_and_size):
        """Same computation composed from the layer-level entry points (one autograd node per layer,
        head as separate ops); kept for tests that cross-check the model-level engine."""
        *rels, size = rels_and_size
        if self.structure in ('GCN', 'GAT'):
            rels = rels[:1]
        index = ops.BatchIndex(adjs, rels, bond_lists=(self.structure == 'GAT'), structure={'Concate': 0, 'Weighted_sum': 1}.get(self.structure, -1))
        return self._forward_composed_index(index, afms, size)

    def _forward_composed_index(self, index, afms, size, seeds=None):
        x, pad_row, layout = self.forward_layers(index, afms, seeds)[-1]
        pad = pad_row if self.structure in ('Weighted_sum', 'GCN') else None
        if self.molfp_mode == 'pool':                                      # models.py:104-106
            g = self.pool1.pooled_sum(index, layout, x, pad, self.graph_layers()[-1])
        else:
            g = ops.readout(index, layout, x, pad, self.molfp_mode, size)  # models.py:108-111
        out, graph_representation = self.head_forward(g, None if seeds is None else seeds[4])
        return out, LazyAtomRep(index, layout, x, pad), graph_representation

    def head_forward(self, g, seed=None):
        """models.py:112-120 on molecule fingerprints g [B, f_last] -> (out, graph_representation): the head's kernels
        (csrc/head2.hip through eagcn_head_forward / eagcn_head_backward, hand-written backward) -- the same stages the model-level
        engine runs behind its read-out.  ``seed``: dropout seed (int, or a 1-element int64 device tensor in graph mode); default: drawn."""
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (self.training and self.dropout > 0) else 0
        mods = {n: getattr(self, n) for n in ('den1', 'den2', 'den3', 'Graph_BN', 'bn_den1', 'bn_den2')}
        return ops.head_forward(mods, g, self.training, seed, self.dropout)

    def head_forward_torch(self, g):
        """The same head as torch ops (BatchNorm1d / relu / dropout modules around the Dense products): kept as the cross-check of
        ``head_forward`` in tests/test_gpu_head.py; no product path calls it."""
        g = self.Graph_BN(g)
        h = F.relu(self.bn_den1(self.den1(g)))
        h = F.dropout(h, p=self.dropout, training=self.training)
        graph_representation = self.den2(h)
        out = self.den3(F.relu(self.bn_den2(graph_representation)))
        return out, graph_representation


class Concate_GCN(EAGCN):
    """``EAGCN(structure='Concate')`` under the name BASELINE.json uses."""

    def __init__(self, *args, **kw):
        kw['structure'] = 'Concate'
        super().__init__(*args, **kw)


class Weighted_GCN(EAGCN):
    """``EAGCN(structure='Weighted_sum')`` under the name BASELINE.json uses."""

    def __init__(self, *args, **kw):
        kw['structure'] = 'Weighted_sum'
        super().__init__(*args, **kw)


def weights_init(m):
    """reference utils.py:702-708 (class-name substring matching, kept verbatim in behaviour)."""
    name = m.__class__.__name__
    if name.find('GraphConv_base') != -1:
        m.weight.data.normal_(0.0, 0.02)
    elif name.find('BatchNorm') != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)
