"""HIP-graph execution of the training step.

The eager engine issues ~60 kernel launches per step; at Tox21 batch sizes the host cannot issue
them as fast as the MI355X retires them.  Every kernel takes its row / tile counts from device
memory and its grid from static capacities, so for a fixed (model, B, N) the launch sequence of the
whole forward and of the whole backward is IDENTICAL from batch to batch: it is captured once into two
HIP graphs and replayed.  What stays eager is only what reads the caller's tensors: the batch-index
kernels (adj / relation tensors) and the packing of ``afms``.  There is no host read-back on the
path; the input-validity counters of batch k are checked (and raised) when batch k+2 is submitted.

Static buffers are sized for ``row_cap`` packed rows (default B*N, which can never overflow).

Everything a batch brings with it is double-buffered (two slots, one captured graph pair each): the
static index, the saved-activation block (whose first region is the packed input), the dropout seeds
and the molecule sizes.  With ``overlap=True`` (inputs already resident in HBM) ALL eager per-batch work
of batch k+1 -- index kernels, packing of ``afms``, seed upload -- runs on a side stream while the GPU is
still executing the backward of batch k; it only has to wait for the backward of batch k-1, the previous
user of its slot.  The main stream then sees nothing but graph launches and the caller's loss.
The captured backward writes the parameter gradients straight into the buffer ``p.grad`` are views of.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from .ops import _index_stream, _ptr, _stream

_RING = 4
# hold the side-stream batch work until the previous step's FORWARD graph has finished: it then runs beside the loss and
# the head / BatchNorm backward (latency chains on a few CUs) instead of beside the layer products (measured round 2:
# 0.5228 -> 0.5175 ms/step at B=256, 1.2386 -> 1.2162 at B=1024; the wave-per-SIMD GEMM is sensitive to co-runners)
_SIDE_AFTER_FWD = os.environ.get('EAGCN_SIDE_AFTER_FWD', '1') == '1'
# EAGCN_SIDE_PLACED=1 (opt-in, measured SLOWER): one captured graph per step has no launch boundary behind the forward to hang an
# event on, so the forward itself bumps a device word behind the read-out (eagcn_model.fwd_signal) and the side stream parks one
# polling wavefront until the previous step has got there (eagcn_stream_wait_counter): the next batch's index scan then runs under
# the head / loss / head-backward kernels instead of beside the persistent plane GEMMs (which it stretches from 150 to 189 us at
# B = 1024, while the one-workgroup `index_offsets` waits 113 us for a CU: profiles/r04_b1024_step_timeline.txt).  Result (round 4,
# two runs each): B = 1024 0.936 -> 0.970 ms, HIV 8.10 -> 8.86, C5 15.97 -> 17.38, Lipo 1.434 -> 1.465, B = 256 unchanged -- the
# index build is 0.14-2.4 ms of side-stream work that needs the WHOLE previous step to hide in; holding it back until that step's
# forward is over puts its tail on the critical path.  Being slowed down by co-runners is cheaper than not overlapping.
# Round 6 (batch hand-off through device flags, the index build's clears in one launch): the placed form wins where the index chain
# FITS under the head: configs[1] 0.378 -> 0.372 ms (the chain no longer meets the forward plane GEMM, which needs empty CUs), and
# still loses where it does not (B = 1024: 0.829 -> 0.870).  `auto` = by the padded rows of the batch.
# Measured again at the end of round 6 (shorter aggregation launches, the signal raised by the head's first launch), same-box A/B on two
# boxes, tools/r6_env_ab.sh: configs[1] 0.374 -> 0.361 and 0.388 -> 0.376 ms WITHOUT the placement -- the unplaced index build again
# hides under the whole step.  Default: off (EAGCN_SIDE_PLACED=0 | auto | 1).
_SIDE_PLACED = os.environ.get('EAGCN_SIDE_PLACED', '0')
_SIDE_PLACED_MAX_ROWS = int(os.environ.get('EAGCN_SIDE_PLACED_MAX_ROWS', '40000'))
# a data-parallel step whose gradient all-reduce cannot be captured into the step graph is an ERROR instead of a (warned)
# fallback to a host-issued collective: bench.py --require-in-graph-allreduce, tools/run_scale.sh
_REQUIRE_IN_GRAPH = os.environ.get('EAGCN_REQUIRE_IN_GRAPH_ALLREDUCE', '0') == '1'
# EAGCN_COMM_IN_GRAPH=0: never try to capture the collective -- the host-issued fallback from the first step on (tests exercise it)
_COMM_IN_GRAPH = os.environ.get('EAGCN_COMM_IN_GRAPH', '1') != '0'
# Capturing a graph that contains collectives, with torch.distributed initialised: the ProcessGroupNCCL watchdog thread polls the
# end events of the EAGER collectives issued before (the warm-up step's all-reduces, the 'dp' loss-scale collective) every 100 ms, and
# an event recorded on the communicator's stream cannot be queried while that stream is part of a capture (hipErrorCapturedEvent:
# "operation not permitted on an event last recorded in a capturing stream") -- the watchdog dies with that exception and takes the
# process with it (SIGABRT; seen in ~1 of 25 runs of tests/dist_multi_check.py: gpurun_out/dist_multi_world1_rc-6.txt).  So the
# eager collectives are allowed to finish AND to be retired by the watchdog before a capture starts: synchronize + this many seconds.
_CAPTURE_DRAIN_S = float(os.environ.get('EAGCN_CAPTURE_DRAIN_S', '0.35'))


# The batch-ready hand-off from the side stream to the step: a device flag that the step's first launch polls
# (eagcn_model.wait_flag) instead of main.wait_event(side event).  hipStreamWaitEvent across two streams costs the waiting stream
# ~40 us per step on ROCm 7.2 even when the event completed long ago (tools/replay_probe.py: 355 us per configs[1] step with
# nothing between the replays, 395 us with an event wait in front of each).  EAGCN_READY_FLAG=0: events.
_READY_FLAG = os.environ.get('EAGCN_READY_FLAG', '1') != '0'
_flag_tested = {}


def _ready_flag_ok(device):
    """Once per device: do the current stream and the index stream really run concurrently?  (Streams that share a hardware queue
    would park the poll in front of its own signal.)  A 50-ms poll on the main stream, the signal on the side stream."""
    if not _READY_FLAG:
        return False
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key not in _flag_tested:
        lib = L.load()
        flag = torch.zeros(1, dtype=torch.int32, device=device)
        torch.cuda.synchronize(device)
        main, side = torch.cuda.current_stream(device), _index_stream(device)
        L.check(lib.eagcn_stream_wait_flag(C.c_void_p(flag.data_ptr()), 0.05, C.c_void_p(main.cuda_stream)), 'eagcn_stream_wait_flag')
        L.check(lib.eagcn_stream_signal_flag(C.c_void_p(flag.data_ptr()), C.c_void_p(side.cuda_stream)), 'eagcn_stream_signal_flag')
        torch.cuda.synchronize(device)
        ok = lib.eagcn_stream_wait_timeouts() == 0
        if not ok:
            import warnings
            lib.eagcn_stream_wait_reset()
            warnings.warn('eagcn_amd: the index stream does not run beside the current stream on this device (shared hardware '
                          'queue?): batch hand-off through stream events instead of a device flag', RuntimeWarning)
        _flag_tested[key] = ok
    return _flag_tested[key]


def _drain_collectives(device):
    """Before capturing a graph in a process with an initialised process group (see _CAPTURE_DRAIN_S)."""
    import torch.distributed as dist
    torch.cuda.synchronize(device)
    if dist.is_available() and dist.is_initialized() and _CAPTURE_DRAIN_S > 0:
        import time
        time.sleep(_CAPTURE_DRAIN_S)


class StaticIndex:
    """BatchIndex-compatible view of the runner's static index buffers (capacities, no host values)."""

    def __init__(self, B, N, channels, device, row_cap, edge_cap=None, rel_c=None, structure=-1):
        K = len(channels)
        self.device, self.B, self.N, self.K = device, B, N, K
        self.channels = list(channels)
        self.ldc = (N + 15) // 16 * 16
        self.T = int(row_cap)
        self.n_tiles = B * ((N + 15) // 16)
        self.n_max = N
        i32 = dict(dtype=torch.int32, device=device)
        self.code = torch.empty((K, B, N, self.ldc), dtype=torch.uint8, device=device)
        blob = torch.zeros(B * N + 3 * B + 2 + L.META_WORDS + 2 * B + 2, **i32)
        rows = torch.zeros(8 * self.T + 5 * self.n_tiles, **i32)
        # bond lists (GAT layers, csrc/gat.hip): capacity in directed bonds; a molecular graph has 2-2.5 per atom
        self.E = int(edge_cap) if edge_cap else min(self.T * N, max(8 * self.T, 1024))
        self._ptrs = torch.zeros(L.bond_ptrs_len(self.T, B), **i32)
        self._edges = torch.zeros(6 * self.E + 2, **i32)       # nbr | tnbr | ecode (u64) | tcode (u64)
        self._blob, self._rows = blob, rows
        o = B * N
        self.nat = blob[o:o + B]
        c = L.Batch()
        c.B, c.N, c.K, c.ldc = B, N, K, self.ldc
        c.T, c.n_max, c.n_tiles = self.T, N, self.n_tiles
        for k in range(K):
            c.channels[k] = self.channels[k]
        base = blob.data_ptr()
        c.code, c.deg_bn, c.nat = self.code.data_ptr(), base, base + 4 * B * N
        c.row0, c.tile0 = base + 4 * (B * N + B), base + 4 * (B * N + 2 * B + 1)
        c.meta = base + 4 * (B * N + 3 * B + 2)
        rb, T = rows.data_ptr(), self.T
        c.row_info, c.tile_info = rb, rb + 16 * T
        ob = rb + 16 * T + 16 * self.n_tiles
        c.row_mol, c.row_loc, c.row_deg, c.row_m, c.tile_mol = ob, ob + 4 * T, ob + 8 * T, ob + 12 * T, ob + 16 * T
        L.set_bond_lists(c, base + 4 * (B * N + 3 * B + 2 + L.META_WORDS), self._ptrs, self._edges, self.E)
        # bond lists + row blocks: built when the library takes a bond-list form of the aggregation for this shape and layer structure
        # (csrc/lagg.hip); the GAT runner sets the flag itself
        self.bond_lists = bool(L.load().eagcn_agg_wants_bond_lists_for(B, N, int(structure)))
        c.build_lists = 1 if self.bond_lists else 0
        # general relation vectors (layers.py:82 with arbitrary channel values): a static 255-row code book per view
        self.relvec = None
        if rel_c is not None:
            self.relvec = [torch.zeros((255, int(cw)), dtype=torch.float32, device=device) for cw in rel_c]
            for k, t in enumerate(self.relvec):
                c.rel_vec[k], c.rel_c[k] = t.data_ptr(), int(t.shape[1])
        self.c = c

    def ref(self):
        return C.byref(self.c)


class GraphRunner:
    """One captured (forward, backward) pair for a fixed (model plan, B, N, channels); with ``training=False``
    a forward-only graph of the eval-mode model (running BatchNorm statistics, no dropout, no backward)."""

    def __init__(self, plan, B, N, channels, device, dropout, row_cap=None, training=True, static_outputs=False,
                 validate='sync', edge_cap=None, rel_c=None):
        lib = L.load()
        self.plan, self.device = plan, device
        self.training = bool(training)
        self.static_outputs = bool(static_outputs)
        self.validate = validate
        self.key = (B, N, tuple(channels))
        structure = plan.specs[0].structure if getattr(plan, 'specs', None) else -1
        self.slots = [StaticIndex(B, N, channels, device, row_cap if row_cap else B * N, edge_cap, rel_c, structure) for _ in range(2)]
        self.rel_vectors = None        # per batch: the code books of a general batch (set by EAGCN._graph_runner)
        self.index = self.slots[0]
        self.graphs = [[None, None, None], [None, None, None]]   # per slot: [forward, backward, whole step (fused loss)]
        self.step_kind = [None, None]
        self.fwd_done = None                                  # main-stream position after the last forward graph
        self.dropout = float(dropout)
        self.seeds_dev = [torch.zeros(8, dtype=torch.int64, device=device) for _ in range(2)]
        self.seeds_host = [torch.zeros(8, dtype=torch.int64).pin_memory() for _ in range(_RING)]
        self.meta_host = [torch.zeros(L.META_WORDS, dtype=torch.int32).pin_memory() for _ in range(_RING)]
        self.meta_event = [None] * _RING
        self.step = 0
        self.cur = 0
        self.t_hint = None             # packed rows of the first batch (_set_row_hint): the size hint of every captured launch
        self.n_in = N                  # padded size of the current batch's tensors (EAGCN(n_bucket=...): may be < N)
        self.generation = 0
        self.size_static = [torch.ones(B, dtype=torch.int64, device=device) for _ in range(2)]
        self.aux = None
        self.dp_group = None
        # optional fork of off-critical-path backward work onto a second stream (graph branches).
        # Measured on MI355X at the Tox21 shape: replay got SLOWER (0.85 vs 0.77 ms/step), so off by default.
        if os.environ.get('EAGCN_AUX_STREAM', '0') == '1':
            self.aux = torch.cuda.Stream(device=self.device)
        self.fwd_sig = torch.zeros(1, dtype=torch.int32, device=device)     # bumped by every executed forward (eagcn_model.fwd_signal)
        # the next batch's index build held back until this step's forward has passed its read-out (module comment at _SIDE_PLACED)
        self.side_placed = _SIDE_PLACED == '1' or (_SIDE_PLACED == 'auto' and B * N <= _SIDE_PLACED_MAX_ROWS)
        # per slot: "this slot's batch is prepared" -- set behind the batch's last preparatory kernel, polled and cleared by the
        # step's first launch (eagcn_model.wait_flag)
        self.ready = torch.zeros(2, dtype=torch.int32, device=device)
        self.use_flag = _ready_flag_ok(device)
        # ... and the other direction: every forward adds 1 to `start_word` in its first launch (eagcn_model.start_signal): when the
        # n-th forward issued on the main stream has STARTED, everything issued before it -- the steps that used the slots last -- is
        # done.  The side stream waits for that count instead of for an event recorded on the main stream between two step graphs.
        # markers[k]: what "the main stream's work up to the step before (k = 1) / before that (k = 0) is done" means:
        # ('count', n) or, without the flags, ('event', ev)
        self.start_word = torch.zeros(1, dtype=torch.int32, device=device)
        self.starts_issued = 0
        self.markers = [None, None]
        self.fwd_issued = 0                                                  # ... and the number of forwards issued so far
        self.cms = [self._cmodel(i) for i in range(2)]
        m = self.cms[0]
        self.saved_bytes = lib.eagcn_model_saved_bytes(self.index.ref(), C.byref(m))
        self.scratch_bytes = lib.eagcn_model_scratch_bytes(self.index.ref(), C.byref(m))
        self.saved = [torch.empty(self.saved_bytes, dtype=torch.uint8, device=device) for _ in range(2)]
        self.scratch = torch.empty(self.scratch_bytes, dtype=torch.uint8, device=device)
        if plan.stats is not None:
            from .parallel import register_stats_buffer
            register_stats_buffer(self.scratch)               # (pieces of it are handed to the sync-BatchNorm hook)
        f32 = dict(dtype=torch.float32, device=device)
        self.out = torch.zeros((B, m.head.nclass), **f32)
        self.graph_rep = torch.zeros((B, m.head.n_den2), **f32)
        self.dout = torch.zeros((B, m.head.nclass), **f32)
        self.dgr = torch.zeros((B, m.head.n_den2), **f32)
        self.dgr_is_zero = True
        self._weight_src = None
        # fused training step (forward + loss + backward in ONE graph): labels per slot, class weights, loss value, DP scale
        self.labels_static = [torch.zeros((B, m.head.nclass), **f32) for _ in range(2)]
        self.weight_static = torch.zeros((m.head.nclass, 2), **f32)
        self.loss_static = [torch.zeros((), **f32) for _ in range(2)]
        self.scale_static = [torch.ones((), **f32) for _ in range(2)]      # per slot: written by the NEXT batch's preparation
        self.comm_in_graph = None if _COMM_IN_GRAPH else False   # gradient all-reduce captured inside the step graph (None: not tried yet)
        n = plan.offsets[-1]
        self.flat_acc = torch.zeros(n, **f32)               # the captured backward writes here; p.grad are views of it
        self.acc_views = plan.grad_views(self.flat_acc)
        self._grads_struct()
        xo, po, ld = C.c_size_t(), C.c_size_t(), C.c_int()
        lib.eagcn_model_atom_rep(self.index.ref(), C.byref(m), C.byref(xo), C.byref(po), C.byref(ld))
        T = self.index.T
        self._xout_views = [sv[xo.value:xo.value + 4 * T * ld.value].view(torch.float32).view(T, ld.value) for sv in self.saved]
        self._pad_views = [sv[po.value:po.value + 4 * ld.value].view(torch.float32) for sv in self.saved]
        self.ptrs = [t.data_ptr() for t in plan._ptr_tensors]

    def nbytes(self):
        """Device memory this runner holds (two index slots, two saved-activation blocks, scratch, gradients)."""
        n = sum(sv.numel() for sv in self.saved) + self.scratch.numel() + 4 * self.flat_acc.numel()
        for sl in self.slots:
            n += sl.code.numel() + 4 * (sl._blob.numel() + sl._rows.numel() + sl._ptrs.numel() + sl._edges.numel())
        return n

    def release(self):
        """Drop the captured graphs and every static buffer (eviction from EAGCN._runners)."""
        torch.cuda.synchronize(self.device)
        self.graphs = [[None, None, None], [None, None, None]]
        self.saved, self.scratch, self.slots, self.index = [], None, [], None
        self._xout_views, self._pad_views = [], []

    def materializer(self):
        """Callable that builds the current slot's top-layer output matrix (atom representations) when the forward skipped it
        (eagcn_model.fuse_readout); None when the forward builds it.  Valid until the slot's next forward, like the views."""
        if not self.cms[self.cur].fuse_readout:
            return None
        cur, gen = self.cur, self.generation

        def run():
            if self.generation - gen >= 2:
                raise L.EagcnHipError('atom representations of a forward that two newer forwards have overwritten')
            L.check(L.load().eagcn_model_atom_rep_materialize(self.slots[cur].ref(), C.byref(self.cms[cur]), _ptr(self.saved[cur]),
                                                              self.saved_bytes, _stream()), 'eagcn_model_atom_rep_materialize')
        return run

    @property
    def xout_view(self):
        return self._xout_views[self.cur]

    @property
    def pad_view(self):
        return self._pad_views[self.cur]

    # -- C descriptors -------------------------------------------------------------------------------
    def _cmodel(self, slot):
        m = L.Model.from_buffer_copy(self.plan.cmodel(self.training, 0, self.dropout))
        sd = self.seeds_dev[slot].data_ptr()
        for l in range(len(self.plan.layers)):
            m.layer[l].seed_dev = sd + 8 * l
        m.head_seed_dev = sd + 8 * 4
        m.input_packed = 1
        if self.use_flag:
            m.wait_flag = self.ready.data_ptr() + 4 * slot
            m.start_signal = self.start_word.data_ptr()
        if self.side_placed:
            m.fwd_signal = self.fwd_sig.data_ptr()
        if self.aux is not None:
            m.aux_stream = self.aux.cuda_stream
        return m

    def _grads_struct(self):
        plan = self.plan
        base = self.flat_acc.data_ptr()

        def gptr(i):
            return base + 4 * plan.offsets[i]
        lg = (L.LayerGrads * len(plan.layers))()
        for l, (layer, (start, ave)) in enumerate(zip(plan.layers, plan.layer_slices)):
            g = lg[l]
            for k in range(layer.K):
                i = start + 6 * k
                g.datt_w[k], g.dself_r[k], g.dW[k] = gptr(i), gptr(i + 1), gptr(i + 2)
                g.dbias[k], g.dgamma[k], g.dbeta[k] = gptr(i + 3), gptr(i + 4), gptr(i + 5)
            g.dave_w = gptr(ave) if ave is not None else None
        hg = L.HeadGrads()
        for j, name in enumerate(('d_den1_w', 'd_den2_w', 'd_den3_w', 'd_gbn_w', 'd_gbn_b', 'd_bn1_w', 'd_bn1_b',
                                  'd_bn2_w', 'd_bn2_b')):
            setattr(hg, name, gptr(plan.head_start + j))
        self.lg, self.hg = lg, hg

    def stale(self):
        """Parameter / buffer storage moved (e.g. .to()): the captured pointers are no longer valid."""
        return [t.data_ptr() for t in self.plan._ptr_tensors] != self.ptrs

    # -- the two launch sequences --------------------------------------------------------------------
    def _call_forward(self):
        lib = L.load()
        if not torch.cuda.is_current_stream_capturing():
            self.fwd_issued += 1                              # (a captured forward counts when its graph is replayed)
            self.starts_issued += 1
        size_ptr = _ptr(self.size_static[self.cur]) if self.plan.molfp else C.c_void_p(0)
        L.check(lib.eagcn_model_forward(self.index.ref(), C.byref(self.cms[self.cur]), C.c_void_p(0), size_ptr,
                                        _ptr(self.saved[self.cur]), self.saved_bytes, _ptr(self.scratch), self.scratch_bytes, _ptr(self.out),
                                        _ptr(self.graph_rep), _stream()), 'eagcn_model_forward')

    def _call_backward(self, with_head=1):
        lib = L.load()
        size_ptr = _ptr(self.size_static[self.cur]) if self.plan.molfp else C.c_void_p(0)
        L.check(lib.eagcn_model_backward_range(self.index.ref(), C.byref(self.cms[self.cur]), size_ptr, _ptr(self.saved[self.cur]),
                                               self.saved_bytes, _ptr(self.scratch), self.scratch_bytes, _ptr(self.dout),
                                               _ptr(self.dgr), self.lg, C.byref(self.hg), with_head, len(self.plan.layers) - 1, 0,
                                               _stream()), 'eagcn_model_backward')

    def _capture(self):
        """Capture both sequences (nothing executes during capture).  Called after the first step ran
        eagerly, which also serves as the warm-up HIP needs before a capture."""
        lib = L.load()
        lib.eagcn_prof_enable(0)
        if self.training and os.environ.get('EAGCN_NO_WARM_BWD', '0') != '1':
            # the backward sequence has never run when the first slot is captured: launch it once eagerly (it only
            # overwrites the gradient buffer and scratch), so that no kernel is launched for the first time inside a
            # capture
            keep = self.flat_acc.clone()              # (gradients of earlier backward passes may be attached to it)
            self._call_backward()
            self.flat_acc.copy_(keep)
        _drain_collectives(self.device)
        fwd = torch.cuda.CUDAGraph()
        # thread_local: other threads (e.g. the RCCL watchdog of torch.distributed) may touch the HIP API during capture
        with torch.cuda.graph(fwd, capture_error_mode='thread_local'):
            self._call_forward()
        bwd = None
        if self.training:
            bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(bwd, capture_error_mode='thread_local'):
                self._call_backward()
        self.graphs[self.cur][0], self.graphs[self.cur][1] = fwd, bwd

    # -- per-step entry points -----------------------------------------------------------------------
    def _check_old_batches(self, force=False, only=None):
        # a timed-out stream-K hand-off (csrc/gemm3.hip) poisons its tile and raises a sticky host-mapped word: the replay
        # loop makes no C call per step, so it is polled here (a plain host read, no synchronisation)
        if L.load().eagcn_stream_wait_timeouts():
            raise L.EagcnHipError('a step waited 2 s for its batch-ready flag (the index stream did not run beside the step: shared '
                                  'hardware queue?): results of that step are invalid; set EAGCN_READY_FLAG=0 (stream events) and '
                                  'call eagcn_stream_wait_reset()')
        if L.load().eagcn_gemm_sk_failed():
            raise L.EagcnHipError('a stream-K GEMM hand-off timed out in an earlier step (a contributor wave was not '
                                  'co-resident with its owner): the gradients of that step are NaN-poisoned; '
                                  'eagcn_gemm_sk_reset_failed() clears the flag')
        # (`only`: the forced wait is for ONE ring slot -- the one about to be reused, the oldest batch in flight.  Waiting for
        #  every slot waits for the NEWEST batch's index build too and pulls the host back to less than one step ahead of the
        #  GPU every time the ring wraps: the next batch's side-stream work was then issued half a step late and the step graph
        #  waited for it, 38 us of idle main stream per 0.4 ms step at configs[1] -- profiles/r06_streams_before.txt)
        for slot in range(_RING):
            ev = self.meta_event[slot]
            if ev is None:
                continue
            forced = force and (only is None or slot == only)
            if not (forced or ev.query()):
                continue
            if forced:
                ev.synchronize()
            meta = self.meta_host[slot].tolist()
            self.meta_event[slot] = None
            if meta[L.META_BAD_ADJ]:
                raise L.EagcnHipError('a previous batch held %d adjacency entries outside {0,1}' % meta[L.META_BAD_ADJ])
            if meta[L.META_BAD_REL]:
                raise L.EagcnHipError('a previous batch held %d bonds whose relation channels are not one-hot'
                                      % meta[L.META_BAD_REL])
            if meta[L.META_OVERFLOW]:
                raise L.EagcnHipError('a previous batch packed %d rows, more than row_cap=%d (it was processed as an '
                                      'empty batch; no memory was overwritten)' % (meta[L.META_OVERFLOW], self.slots[0].T))
            if meta[L.META_EDGE_OVERFLOW]:
                raise L.EagcnHipError('a previous batch held %d directed bonds, more than edge_cap=%d (it was processed as '
                                      'an empty batch; no memory was overwritten)' % (meta[L.META_EDGE_OVERFLOW], self.slots[0].E))

    def _prepare(self, adj, rels, afm, size, seed, overlap=False, bonds=None, labels=None, dp_scale=None):
        """Everything of a step that reads the caller's tensors (index build, packed input, seeds, sizes, labels of a fused
        step), on the side stream when `overlap`; afterwards the main stream is ordered behind it."""
        lib = L.load()
        self._check_old_batches()
        self.cur = cur = self.step % 2
        idx = self.index = self.slots[cur]
        main = torch.cuda.current_stream(self.device)
        slot = self.step % _RING
        if self.meta_event[slot] is not None:                 # ring wrapped: this slot must be consumed first
            self._check_old_batches(force=True, only=slot)
        # main-stream position now = after the backward of the previous step; the position recorded at the
        # PREVIOUS forward entry = after the backward of the step before it, the last user of this slot
        if self.use_flag:
            # the slot's last user is the step before the previous one: it is done once the forward issued LAST (the previous step's)
            # has started -- the count of forwards issued up to now
            slot_free = ('count', self.starts_issued) if self.starts_issued >= 2 else None
        else:
            # main-stream position now = after the backward of the previous step; the position recorded at the PREVIOUS forward
            # entry = after the backward of the step before it, the last user of this slot
            ev = torch.cuda.Event()
            ev.record(main)
            slot_free = self.markers[1]
            self.markers = [self.markers[1], ('event', ev)]
        if overlap:
            side = _index_stream(self.device)
            if slot_free is not None and slot_free[0] == 'event':
                side.wait_event(slot_free[1])
            elif slot_free is not None:
                L.check(lib.eagcn_stream_wait_counter(C.c_void_p(self.start_word.data_ptr()), slot_free[1] & 0xFFFFFFFF,
                                                      C.c_void_p(side.cuda_stream)), 'eagcn_stream_wait_counter')
            # ... and it is held back until the FORWARD of the previous step has finished: from there on the main
            # stream runs the caller's loss and the head's backward -- short kernels on a few CUs -- under which
            # the HBM-streaming index scan costs nothing (at the start of a step it would compete with the
            # layer GEMMs and aggregations)
            if self.fwd_done is not None and _SIDE_AFTER_FWD:
                side.wait_event(self.fwd_done)
            elif self.side_placed and self.fwd_issued > 0:
                # fused steps: until the forward of the step issued last has passed its read-out (it is already enqueued on the
                # main stream, and the main stream never waits for anything issued behind this point of the side stream)
                L.check(lib.eagcn_stream_wait_counter(C.c_void_p(self.fwd_sig.data_ptr()), self.fwd_issued & 0xFFFFFFFF,
                                                      C.c_void_p(side.cuda_stream)), 'eagcn_stream_wait_counter')
            for t in (afm, adj, size, labels) + tuple(rels or ()) + tuple(bonds or ()):
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(side)                     # read on the side stream after this call returns
        else:
            side = main
        istream = C.c_void_p(side.cuda_stream)
        idx.c.n_logical = int(self.n_in) if self.n_in != idx.N else 0     # row stride of the caller's tensors / logical N
        # ---- everything that reads the caller's tensors: index, packed input, seeds, sizes -----------------
        if bonds is None:
            rel_ptrs = (C.c_void_p * idx.K)(*[r.data_ptr() for r in rels])
            L.check(lib.eagcn_index_build(_ptr(adj), rel_ptrs, idx.ref(), C.c_void_p(self.meta_host[slot].data_ptr()),
                                          istream), 'eagcn_index_build')
        else:                                                 # compact batch: O(bonds) input instead of the dense tensors
            bm, bi, bj, bc = bonds
            L.check(lib.eagcn_index_from_bonds(_ptr(bm), _ptr(bi), _ptr(bj), _ptr(bc), bm.numel(), idx.ref(),
                                               C.c_void_p(self.meta_host[slot].data_ptr()), istream),
                    'eagcn_index_from_bonds')
        L.check(lib.eagcn_index_rows(idx.ref(), istream), 'eagcn_index_rows')
        L.check(lib.eagcn_model_pack_input(idx.ref(), C.byref(self.cms[cur]), _ptr(afm), _ptr(self.saved[cur]),
                                           self.saved_bytes, istream), 'eagcn_model_pack_input')
        sh = self.seeds_host[slot]                            # pinned staging, one per ring slot (async copy source)
        sn = sh.numpy()                                       # same derivation as ModelPlan.cmodel
        for l in range(4):
            sn[l] = (seed + 7919 * (l + 1)) & (2 ** 63 - 1)
        sn[4] = (seed + 0x51ED27) & (2 ** 63 - 1)
        with torch.cuda.stream(side):
            self.seeds_dev[cur].copy_(sh, non_blocking=True)
            if self.plan.molfp:
                self.size_static[cur].copy_(size, non_blocking=True)
            if labels is not None:
                self.labels_static[cur].copy_(labels.reshape(self.labels_static[cur].shape), non_blocking=True)
            if dp_scale is not None:
                if isinstance(dp_scale, str):
                    # c_r = world * n_r / sum_r n_r (parallel.dp_loss_scale) from THIS batch's labels: a 1-element collective
                    # that rides with the batch's preparatory work instead of sitting in front of the step graph
                    from .parallel import dp_loss_scale
                    dp_scale = dp_loss_scale(self.labels_static[cur], self.dp_group)
                self.scale_static[cur].copy_(dp_scale, non_blocking=True)
            if idx.relvec is not None:                       # code books of this batch (rows beyond them are never referenced)
                for t, v in zip(idx.relvec, self.rel_vectors):
                    t[:v.shape[0]].copy_(v, non_blocking=True)
        ev = torch.cuda.Event()           # meta_host[slot] is valid and seeds_host[slot] is free again after this point
        ev.record(side)
        self.meta_event[slot] = ev
        if self.validate == 'sync':
            # wait for THIS batch's index kernels (they only depend on the slot's previous user, not on the main
            # stream's backlog) and raise on invalid input / row_cap overflow before anything consumes the index
            ev.synchronize()
            meta = self.meta_host[slot].tolist()
            self.meta_event[slot] = None
            bad = None
            if meta[L.META_BAD_ADJ]:
                bad = ('%d bonds are out of range or listed twice' if bonds is not None else
                       'adjs holds %d entries outside {0,1}') % meta[L.META_BAD_ADJ]
            elif meta[L.META_BAD_REL]:
                bad = '%d bonded (i,j,view) positions are not one-hot over the relation channels' % meta[L.META_BAD_REL]
            elif meta[L.META_OVERFLOW]:
                bad = 'the batch packs %d rows, more than row_cap=%d' % (meta[L.META_OVERFLOW], idx.T)
            elif meta[L.META_EDGE_OVERFLOW]:
                bad = 'the batch holds %d directed bonds, more than edge_cap=%d' % (meta[L.META_EDGE_OVERFLOW], idx.E)
            if bad:
                if overlap:
                    main.wait_stream(side)
                raise L.EagcnHipError(bad)
        if self.use_flag:
            # (behind everything above, torch's copies included; the step's first launch waits for it on the device)
            L.check(lib.eagcn_stream_signal_flag(C.c_void_p(self.ready.data_ptr() + 4 * cur), istream), 'eagcn_stream_signal_flag')
        elif overlap:
            done = torch.cuda.Event()
            done.record(side)
            main.wait_event(done)
        self.step += 1
        self.generation += 1
        if self.training:
            self.plan.nbt_pending += 1    # num_batches_tracked: counted on the host, written by ModelPlan.flush_nbt()
        return main

    def _set_row_hint(self):
        """Before a slot's sequences are captured: tell the library how many packed rows batches of this shape hold (eagcn_batch.t_hint).
        The captured launches are fixed for every later batch of the shape; the buffers are sized for row_cap (default B * N, 7x
        the rows of a Tox21 batch), and the size-dependent kernel choices (the plane GEMM's tile shape, the aggregation form, the
        grid of the bond-list aggregation, the slab layout of the layer scratch) should follow the rows batches of this shape really
        have.  ONE hint per runner -- the rows of the FIRST batch it sees -- for both slots and every kind of captured graph, fixed from
        then on: a slot that captured under another batch's count would freeze other kernels than its twin (even and odd steps would
        differ in timing and rounding).  One host wait, once per runner."""
        if self.t_hint is None:
            slot = (self.step - 1) % _RING
            ev = self.meta_event[slot]
            if ev is not None:
                ev.synchronize()
            t = int(self.meta_host[slot][L.META_T])
            self.t_hint = max(1, min(t, self.index.T)) if t > 0 else 0
        for sl in self.slots:
            sl.c.t_hint = self.t_hint

    def _reset_ready(self):
        """A step failed between its batch-ready signal and the launch that consumes it: no flag may stay set."""
        if self.use_flag:
            torch.cuda.synchronize(self.device)
            self.ready.zero_()
            v = self.starts_issued & 0xFFFFFFFF
            self.start_word.fill_(v - 2 ** 32 if v >= 2 ** 31 else v)  # (an int32 tensor: the count as the device word holds it)
            torch.cuda.synchronize(self.device)

    def forward(self, adj, rels, afm, size, seed, overlap=False, bonds=None):
        main = self._prepare(adj, rels, afm, size, seed, overlap, bonds)
        cur = self.cur
        try:
            if self.graphs[cur][0] is None:
                self._set_row_hint()
                self._call_forward()                          # first use of a slot: eager (and the capture warm-up)
                self._capture()
            else:
                self.graphs[cur][0].replay()
                self.fwd_issued += 1
                self.starts_issued += 1
        except BaseException:
            self._reset_ready()
            raise
        if overlap:
            self.fwd_done = torch.cuda.Event()
            self.fwd_done.record(main)
        return self.generation

    def outputs(self):
        """(out, graph_representation) of the last forward.  By default fresh tensors, as the reference returns
        (code that collects predictions / embeddings over several batches, train.py:279-285, keeps working);
        with static_outputs=True views of the static buffers that the next forward of this shape overwrites."""
        if self.static_outputs:
            return self.out.detach(), self.graph_rep.detach()
        return self.out.clone(), self.graph_rep.clone()

    def backward(self, dout, dgr, generation):
        if generation != self.generation:
            raise L.EagcnHipError('graph mode keeps ONE forward in flight: backward() of an older forward was '
                                  'called after a newer training forward overwrote the saved activations')
        if dout.data_ptr() != self.dout.data_ptr():          # the fused losses write straight into self.dout
            self.dout.copy_(dout)
        if dgr is not None:
            self.dgr.copy_(dgr)
            self.dgr_is_zero = False
        elif not self.dgr_is_zero:
            self.dgr.zero_()
            self.dgr_is_zero = True
        keep, grads = self._before_grads()
        if self.graphs[self.cur][1] is None:
            self._call_backward()
        else:
            self.graphs[self.cur][1].replay()
        self._attach_grads(keep, grads)

    def _before_grads(self):
        # (frozen buffers that ride in plan.params -- the constant attention weights of Vanilla_GCN.hot_params -- never get a
        #  .grad: optimizer.zero_grad would not clear it and every later step would take the accumulate path below)
        grads = [p.grad if p.requires_grad else None for p in self.plan.params]
        # the captured backward OVERWRITES flat_acc (the storage p.grad are views of); if gradients of an earlier
        # backward are still attached (accumulation across backward calls) keep them and add afterwards
        keep = None
        if any(g is v for g, v in zip(grads, self.acc_views)):
            keep = self.flat_acc.clone()
        return keep, grads

    def _attach_grads(self, keep, grads):
        params, views = self.plan.params, self.acc_views
        if keep is None and all(g is None for g in grads):
            for p, v in zip(params, views):
                if p.requires_grad:
                    p.grad = v
            return
        kept = self.plan.grad_views(keep) if keep is not None else None
        for i, (p, g, v) in enumerate(zip(params, grads, views)):
            if not p.requires_grad:
                continue
            if g is None:
                p.grad = v
            elif g is v:
                v.add_(kept[i])
            else:
                g.add_(v)                                     # a gradient tensor of the caller's: accumulate into it

    # -- fused training step: forward + loss + backward as ONE graph launch -----------------------------
    def _call_forward_step(self, kind, scaled):
        """Forward, loss and the head's backward (eagcn_model_forward_step: the head's eight stages as one launch); the layers'
        backward follows through _call_backward(with_head=0)."""
        lib = L.load()
        if not torch.cuda.is_current_stream_capturing():
            self.fwd_issued += 1
            self.starts_issued += 1
        cur = self.cur
        sl = L.StepLoss()
        sl.kind = 0 if kind == 'bce' else 1
        sl.labels = self.labels_static[cur].data_ptr()
        sl.class_weight = self.weight_static.data_ptr()
        sl.loss = self.loss_static[cur].data_ptr()
        sl.scale = self.scale_static[cur].data_ptr() if scaled else None       # data-parallel global normalisation (parallel.dp_loss_scale)
        sl.dout = self.dout.data_ptr()
        size_ptr = _ptr(self.size_static[cur]) if self.plan.molfp else C.c_void_p(0)
        L.check(lib.eagcn_model_forward_step(self.index.ref(), C.byref(self.cms[cur]), C.c_void_p(0), size_ptr, _ptr(self.saved[cur]),
                                             self.saved_bytes, _ptr(self.scratch), self.scratch_bytes, _ptr(self.out), _ptr(self.graph_rep),
                                             C.byref(sl), _ptr(self.dgr), C.byref(self.hg), _stream()), 'eagcn_model_forward_step')

    def _call_backward_comm(self, comm, with_head=1):
        """The backward with the gradient average inside: head + upper layers, then the all-reduce of their bucket of the flat
        gradient buffer is STARTED (asynchronously: under capture a branch of the graph), the first layer's backward runs
        beside it, and its own (small) bucket follows.  One bucket when the model has a single layer."""
        lib = L.load()
        nl = len(self.plan.layers)
        if nl < 2:
            self._call_backward(with_head)
            comm.start(self.flat_acc).wait()
            return
        size_ptr = _ptr(self.size_static[self.cur]) if self.plan.molfp else C.c_void_p(0)

        def part(with_head, hi, lo):
            L.check(lib.eagcn_model_backward_range(self.index.ref(), C.byref(self.cms[self.cur]), size_ptr, _ptr(self.saved[self.cur]),
                                                   self.saved_bytes, _ptr(self.scratch), self.scratch_bytes, _ptr(self.dout),
                                                   _ptr(self.dgr), self.lg, C.byref(self.hg), with_head, hi, lo, _stream()),
                    'eagcn_model_backward_range')
        part(with_head, nl - 1, 1)
        cut = self.plan.offsets[self.plan.layer_slices[1][0]]      # first gradient of the second layer: [0, cut) = layer 1
        upper = comm.start(self.flat_acc[cut:])
        part(0, 0, 0)
        upper.wait()
        comm.start(self.flat_acc[:cut]).wait()

    def train_step(self, adj, rels, afm, size, seed, labels, kind, weight=None, scale=None, overlap=False, bonds=None,
                   comm=None, optimizer=None):
        """forward -> fused loss (csrc/loss.hip) -> backward of one batch as a single captured graph: no launch boundary
        between the three, one host call per step.  Returns the loss (device scalar); out / graph_representation are read
        with outputs(), the parameter gradients are attached exactly as backward() does.  `comm` (a GradientAllReducer with
        more than one rank behind it): the gradient average is captured into the same graph (_call_backward_comm); should the
        capture of the collective fail on this stack, the runner falls back to one host-issued all-reduce after the replay."""
        if not self.training:
            raise L.EagcnHipError('train_step needs a training-mode runner')
        labels = labels.to(device=self.device, dtype=torch.float32)
        if labels.numel() != self.labels_static[0].numel():
            raise L.EagcnHipError('train_step: %d labels for logits %s' % (labels.numel(), tuple(self.out.shape)))
        if kind == 'bce':
            w = weight.to(device=self.device, dtype=torch.float32)
            if tuple(w.shape) != tuple(self.weight_static.shape):
                raise L.EagcnHipError('bce loss: weight %s for %d tasks' % (tuple(w.shape), self.weight_static.shape[0]))
            tag = (id(w), w._version)
            if self._weight_src != tag:                       # (class weights change once per run, not per step)
                self.weight_static.copy_(w)
                self._weight_src = tag
        if comm is not None and not comm.active():
            comm = None
        self.dp_group = comm.group if comm is not None else None
        if not self.dgr_is_zero:
            self.dgr.zero_()
            self.dgr_is_zero = True
        keep, grads = self._before_grads()
        if optimizer is not None and (keep is not None or any(g is not None for g in grads)):
            raise L.EagcnHipError('train_step(optimizer=...): the update is part of the step graph and uses this step\'s gradients; '
                                  'call optimizer.zero_grad() (set_to_none) before the step')
        self._prepare(adj, rels, afm, size, seed, overlap, bonds, labels=labels, dp_scale=scale)
        cur = self.cur
        in_graph = comm is not None and self.comm_in_graph is not False
        # `optimizer` (eagcn_amd.optim.FlatAdam): the parameter update is the last launch of the captured step (its hyper-parameters
        # and step count live in device memory)
        key = (kind, scale is not None, in_graph, id(optimizer) if optimizer is not None else None)

        def update():
            if optimizer is not None:
                optimizer.launch(self.flat_acc)
        try:
            first_eager = self.graphs[cur][2] is None or self.step_kind[cur] != key
            if first_eager:
                self._set_row_hint()
                self._call_forward_step(key[0], key[1])           # eager (first use of the slot / of this loss): the warm-up
                if in_graph:
                    self._call_backward_comm(comm, 0)
                else:
                    self._call_backward(0)
                    if comm is not None:
                        # host-issued average of the flat buffer itself (the .grad views are attached only below: after zero_grad they are
                        # None here and GradientAllReducer.__call__ would find nothing to reduce); the update below needs the average
                        comm.start(self.flat_acc).wait()
                update()
                _drain_collectives(self.device)
                L.load().eagcn_prof_enable(0)
                g = torch.cuda.CUDAGraph()
                captured, failure = True, None
                try:
                    with torch.cuda.graph(g, capture_error_mode='thread_local'):
                        self._call_forward_step(key[0], key[1])
                        if in_graph:
                            self._call_backward_comm(comm, 0)
                            update()
                        else:
                            self._call_backward(0)
                            if comm is None:
                                update()
                except L.EagcnHipError:
                    if in_graph:
                        comm.agree(False)                         # (the other ranks are waiting in agree() below: fail with them, not hang them)
                    raise                                         # one of OUR launches failed: never hidden behind a re-capture
                except Exception as e:                            # noqa: BLE001 -- the capture of the collective failed on this stack
                    if not in_graph:
                        raise
                    captured, failure = False, e
                    if optimizer is not None:
                        optimizer.reset_ticket()                  # (an aborted capture must not leave the update's last-workgroup ticket half counted)
                if in_graph:
                    # every rank must replay the SAME collective sequence: a rank whose capture failed while the others replay
                    # in-graph all-reduces would hang them.  One MIN-reduction of the success flag decides for all ranks.
                    ok = comm.agree(captured)
                    if ok:
                        self.comm_in_graph = True
                    elif _REQUIRE_IN_GRAPH:
                        raise L.EagcnHipError('the gradient all-reduce could not be captured into the step graph on every rank '
                                              '(EAGCN_REQUIRE_IN_GRAPH_ALLREDUCE=1): %r' % (failure,))
                if in_graph and not self.comm_in_graph:
                    import warnings
                    warnings.warn('eagcn_amd: the gradient all-reduce could not be captured into the step graph (%r); falling back '
                                  'to one host-issued all-reduce behind every replay' % (failure,), RuntimeWarning)
                    # the collective could not be captured: step graph without it, host-issued all-reduce behind every replay
                    _drain_collectives(self.device)
                    self.comm_in_graph, in_graph = False, False
                    key = (kind, scale is not None, False, key[3])
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode='thread_local'):
                        self._call_forward_step(key[0], key[1])
                        self._call_backward(0)
                    # (the eager step above left the AVERAGED gradients in the flat buffer; the host-issued average below is then
                    #  the identity on values that are equal on every rank)
                self.graphs[cur][2], self.step_kind[cur] = g, key
            else:
                self.graphs[cur][2].replay()
                self.fwd_issued += 1
                self.starts_issued += 1
        except BaseException:
            self._reset_ready()
            raise
        self.fwd_done = None            # (no launch boundary after the forward any more: the next index build only waits
                                        #  for its slot and runs under this step's kernels)
        self._attach_grads(keep, grads)
        if comm is not None and not in_graph and not first_eager:
            comm()                                            # one in-place average of the flat buffer, issued by the host
            update()                                          # ... and the update behind it (outside the graph in this fallback)
        return self.loss_static[cur].detach() if self.static_outputs else self.loss_static[cur].clone()


class _GraphFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, adj, rels, afm, size, seed, overlap, trigger, bonds=None):
        ctx.set_materialize_grads(False)
        if ctx.needs_input_grad[3]:
            raise L.EagcnHipError('graph mode does not produce d/d(afms); use EAGCN.forward_composed for input gradients')
        ctx.runner = runner
        ctx.generation = runner.forward(adj, rels, afm, size, seed, overlap, bonds)
        return runner.outputs()

    @staticmethod
    def backward(ctx, dout, dgr):
        if dout is None:
            dout = torch.zeros_like(ctx.runner.out)
        ctx.runner.backward(dout.contiguous(), None if dgr is None else dgr.contiguous(), ctx.generation)
        return (None,) * 9


def graph_forward(runner, adj, rels, afm, size, seed, overlap=False, bonds=None):
    if not runner.training:               # eval: forward-only graph, nothing to differentiate
        runner.forward(adj, rels, afm, size, seed, overlap, bonds)
        return runner.outputs()
    plan = runner.plan
    if plan.trigger is None or plan.trigger.device != afm.device:
        plan.trigger = torch.zeros((), dtype=torch.float32, device=afm.device, requires_grad=True)
    out, graph_rep = _GraphFn.apply(runner, adj, rels, afm, size, seed, overlap, plan.trigger, bonds)
    out._eagcn_grad_slot = runner.dout    # hints for eagcn_amd.losses: where d(loss)/d(out) is consumed ...
    out._eagcn_step = (runner, runner.generation)      # ... and which captured backward belongs to this forward
    return out, graph_rep
