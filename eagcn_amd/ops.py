"""Host side of the HIP hot path: batch index + autograd Functions over the C ABI.

PyTorch is used here for device memory (caching allocator), the current HIP stream and autograd
bookkeeping only; every computation is a call into libeagcn_hip.so.  There is no eager / CPU
fallback: tensors that are not fp32 CUDA(HIP) tensors raise.
"""
import ctypes as C
import os

import torch

from . import _lib as L


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _need_cuda_f32(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise L.EagcnHipError('%s must be a tensor on the MI355X (got %s): eagcn_amd has no CPU path'
                              % (name, getattr(t, 'device', type(t))))
    if t.dtype != torch.float32:
        raise L.EagcnHipError('%s must be float32 (got %s)' % (name, t.dtype))
    return t.contiguous()


_pinned_meta = {}
_index_streams = {}


def _host_meta(device):
    key = device.index
    if key not in _pinned_meta:
        _pinned_meta[key] = torch.zeros(L.META_WORDS, dtype=torch.int32).pin_memory()
    return _pinned_meta[key]


def _index_stream(device):
    """Side stream for the batch-index kernels.  The host has to read the packed row count before it
    can size the rest of the step; waiting on the MAIN stream would drain everything queued there (the
    previous step's backward).  On a side stream the wait covers the two index kernels only, so the
    host keeps running one step ahead of the GPU."""
    key = device.index
    if key not in _index_streams:
        _index_streams[key] = torch.cuda.Stream(device=device, priority=int(os.environ.get('EAGCN_SIDE_PRIORITY', '0')))
    return _index_streams[key]


class ColLayout:
    """Column layout of a packed activation matrix: exact widths + padded widths per segment."""

    def __init__(self, widths, pads):
        self.widths = [int(w) for w in widths]
        self.pads = [int(p) for p in pads]
        self.ld = sum(self.pads)
        self.width = sum(self.widths)
        self.c = L.make_layout(self.widths, self.pads)

    @staticmethod
    def single(width, mult=4):
        return ColLayout([width], [(int(width) + mult - 1) // mult * mult])

    def __repr__(self):
        return 'ColLayout(%s -> %s)' % (self.widths, self.pads)


class BatchIndex:
    """Compact description of one collated batch (reference utils.py:575-640 tensors).

    Built once per batch and shared by every layer (the reference rebuilds its masks in every
    layer, layers.py:294-304).  Construction runs two small kernels and one 32-byte D2H copy that
    the host waits for: it returns the number of packed rows (buffer sizes / grid sizes) and the
    input-validity counters -- non-binary adjacency or non-one-hot relation channels raise here.
    """

    @classmethod
    def from_bonds(cls, B, N, channels, bond_mol, bond_i, bond_j, bond_code, row_cap=None, rel_vectors=None, bond_lists=False):
        """Index of a COMPACT batch: directed bonds as int32 device vectors bond_mol / bond_i /
        bond_j [E] plus their type per view bond_code [E,K] (uint8).  Equivalent to BatchIndex(adj, rels) on
        the dense tensors the reference's collate would build from the same molecules, without ever
        materialising them (O(E) input bytes instead of 4*(1+sum C_k)*B*N*N)."""
        self = cls.__new__(cls)
        self.rel_vectors = None
        self.bond_lists = bool(bond_lists)
        if rel_vectors is not None:                   # general relation tensors: a code stands for a channel vector
            if len(rel_vectors) != len(channels):
                raise L.EagcnHipError('rel_vectors needs one table per view')
            self.rel_vectors = []
            for k, v in enumerate(rel_vectors):
                v = _need_cuda_f32(v, 'rel_vectors[%d]' % k).contiguous()
                if v.dim() != 2 or v.shape[0] != channels[k] or not 1 <= v.shape[0] <= 255:
                    raise L.EagcnHipError('rel_vectors[%d] must be [%d codes <= 255, C], got %s' % (k, channels[k], tuple(v.shape)))
                self.rel_vectors.append(v)
        self._build(None, None, B, N, list(channels), bond_mol.device, False, row_cap,
                    bonds=(bond_mol, bond_i, bond_j, bond_code))
        return self

    def __init__(self, adj, rels, overlap=False, row_cap=None, bond_lists=False, structure=-1):
        """structure: EAGCN_STRUCT_* of the layers the index will serve (-1: not known): whether the library takes a bond-list form
        of the aggregation -- and the index has to carry bond lists and row blocks -- depends on it (eagcn_agg_wants_bond_lists_for).
        row_cap: size every packed buffer / grid for `row_cap` rows instead of the exact packed row
        count (kernels read the exact count from device memory); used by tests and by graph mode.
        overlap=True: the inputs are already materialised in HBM (prefetched batches), so the index
        kernels may run on a side stream WITHOUT waiting for the main stream's backlog; the host then
        only waits for those two kernels and keeps running one step ahead of the GPU.  Default False:
        the side stream first waits for everything queued on the current stream (always safe)."""
        self.structure = int(structure)
        lib = L.load()
        adj = _need_cuda_f32(adj, 'adjs')
        rels = [_need_cuda_f32(r, 'relation tensor %d' % i) for i, r in enumerate(rels)]
        if adj.dim() != 3 or adj.shape[1] != adj.shape[2]:
            raise L.EagcnHipError('adjs must be [B,N,N], got %s' % (tuple(adj.shape),))
        B, N, _ = adj.shape
        K = len(rels)
        if not 1 <= K <= L.MAX_VIEWS:
            raise L.EagcnHipError('between 1 and %d relation tensors supported, got %d' % (L.MAX_VIEWS, K))
        for i, r in enumerate(rels):
            if r.dim() != 4 or r.shape[0] != B or r.shape[2] != N or r.shape[3] != N:
                raise L.EagcnHipError('relation tensor %d must be [B,C,N,N] = [%d,C,%d,%d], got %s'
                                      % (i, B, N, N, tuple(r.shape)))
        self.rel_vectors = None
        self.bond_lists = bool(bond_lists)
        self._build(adj, rels, B, N, [int(r.shape[1]) for r in rels], adj.device, overlap, row_cap)

    def _build(self, adj, rels, B, N, channels, dev, overlap, row_cap, bonds=None):
        lib = L.load()
        K = len(channels)
        self.device, self.B, self.N, self.K = dev, B, N, K
        self.channels = channels
        self.ldc = (N + 15) // 16 * 16
        i32 = dict(dtype=torch.int32, device=dev)
        main = torch.cuda.current_stream(dev)
        side = _index_stream(dev)
        # buffers written on the side stream are allocated on it (the caching allocator recycles blocks
        # per stream: a block freed on `main` may still be read by kernels queued there)
        with torch.cuda.stream(side):
            self.code = torch.empty((K, B, N, self.ldc), dtype=torch.uint8, device=dev)
            # one allocation for the small int32 arrays: [deg_bn B*N | nat B | row0 B+1 | tile0 B+1 | meta]
            blob = torch.empty(B * N + 3 * B + 2 + L.META_WORDS + 2 * B + 2, **i32)
        o = 0
        self.deg_bn = blob[o:o + B * N]; o += B * N
        self.nat = blob[o:o + B]; o += B
        self.row0 = blob[o:o + B + 1]; o += B + 1
        self.tile0 = blob[o:o + B + 1]; o += B + 1
        self.meta = blob[o:o + L.META_WORDS]
        self._blob = blob
        c = L.Batch()
        c.B, c.N, c.K, c.ldc = B, N, K, self.ldc
        for k in range(K):
            c.channels[k] = self.channels[k]
        base = blob.data_ptr()
        c.code, c.deg_bn, c.nat = self.code.data_ptr(), base, base + 4 * B * N
        c.row0, c.tile0 = base + 4 * (B * N + B), base + 4 * (B * N + 2 * B + 1)
        c.meta = base + 4 * (B * N + 3 * B + 2)
        for k, v in enumerate(getattr(self, 'rel_vectors', None) or ()):
            c.rel_vec[k], c.rel_c[k] = v.data_ptr(), int(v.shape[1])
        c.ecnt = base + 4 * (B * N + 3 * B + 2 + L.META_WORDS)
        c.edge0 = c.ecnt + 4 * B
        c.E = 1 << 30                     # (exact-size edge arrays are allocated once the bond count is known)
        host = _host_meta(dev)
        if not overlap:
            side.wait_stream(main)        # inputs may have been produced by work still queued on `main`
        sptr = C.c_void_p(side.cuda_stream)
        if bonds is None:
            rel_ptrs = (C.c_void_p * K)(*[r.data_ptr() for r in rels])
            L.check(lib.eagcn_index_build(_ptr(adj), rel_ptrs, C.byref(c), C.c_void_p(host.data_ptr()), sptr),
                    'eagcn_index_build')
        else:
            bm, bi, bj, bc = bonds
            for t, dt, name in ((bm, torch.int32, 'bond_mol'), (bi, torch.int32, 'bond_i'), (bj, torch.int32, 'bond_j'),
                                (bc, torch.uint8, 'bond_code')):
                if not t.is_cuda or t.dtype != dt or not t.is_contiguous():
                    raise L.EagcnHipError('%s must be a contiguous %s device tensor' % (name, dt))
            E = bm.numel()
            if bi.numel() != E or bj.numel() != E or tuple(bc.shape) != (E, K):
                raise L.EagcnHipError('bond arrays disagree: %d / %d / %d bonds, codes %s' % (E, bi.numel(), bj.numel(), tuple(bc.shape)))
            L.check(lib.eagcn_index_from_bonds(_ptr(bm), _ptr(bi), _ptr(bj), _ptr(bc), E, C.byref(c),
                                               C.c_void_p(host.data_ptr()), sptr), 'eagcn_index_from_bonds')
        side.synchronize()                # waits for the index kernels only, not for the main stream's backlog
        meta = host.tolist()
        if meta[L.META_BAD_ADJ] and bonds is not None:
            raise L.EagcnHipError('%d bonds are out of range or listed twice' % meta[L.META_BAD_ADJ])
        if meta[L.META_BAD_ADJ]:
            raise L.EagcnHipError('adjs holds %d entries outside {0,1}: the hot path requires a 0/1 '
                                  'adjacency (reference neural_fp.py:85,109-110)' % meta[L.META_BAD_ADJ])
        if meta[L.META_BAD_REL]:
            raise L.EagcnHipError('%d bonded (i,j,view) positions are not one-hot over the relation channels '
                                  '(reference neural_fp.py:111-120); build the model with relations=\'general\' for relation '
                                  'tensors with arbitrary channel values (layers.py:82)' % meta[L.META_BAD_REL])
        self.T, self.n_max, self.n_tiles = meta[L.META_T], meta[L.META_NMAX], meta[L.META_NTILES]
        self.n_edges = meta[L.META_NEDGE]
        self.rows = self.T                                   # exact packed row count
        if row_cap is not None:
            if row_cap < self.T:
                raise L.EagcnHipError('row_cap %d is smaller than the packed row count %d' % (row_cap, self.T))
            self.T = int(row_cap)                            # capacity from here on
            self.n_tiles = B * ((N + 15) // 16)
        T = self.T
        main.wait_stream(side)            # everything below (and every consumer) runs on `main`
        # [row_info 4T | tile_info 4*n_tiles (both 16-byte aligned) | row_mol | row_loc | row_deg | row_m(f32) | tile_mol]
        nt = self.n_tiles
        rows = torch.empty(8 * T + 5 * nt, **i32)
        o = 4 * T + 4 * nt
        self.row_mol, self.row_loc, self.row_deg = rows[o:o + T], rows[o + T:o + 2 * T], rows[o + 2 * T:o + 3 * T]
        self.row_m = rows[o + 3 * T:o + 4 * T].view(torch.float32)
        self.tile_mol = rows[o + 4 * T:]
        self._rows = rows
        c.T, c.n_max, c.n_tiles = T, self.n_max, self.n_tiles
        c.t_hint = int(self.rows)                            # (exact here; with a row_cap the buffers are larger than the batch)
        rb = rows.data_ptr()
        c.row_info, c.tile_info = rb, rb + 16 * T
        ob = rb + 4 * o
        c.row_mol, c.row_loc, c.row_deg, c.row_m, c.tile_mol = ob, ob + 4 * T, ob + 8 * T, ob + 12 * T, ob + 16 * T
        self.E = max(int(self.n_edges), 1)
        self._ptrs = torch.zeros(L.bond_ptrs_len(T, B), **i32)
        self._edges = torch.empty(6 * self.E + 2, **i32)
        L.set_bond_lists(c, base + 4 * (B * N + 3 * B + 2 + L.META_WORDS), self._ptrs, self._edges, self.E)
        # bond lists: the GAT layers, and the bond-list form of the aggregation where the library picks it (csrc/lagg.hip)
        if lib.eagcn_agg_wants_bond_lists_for(B, N, int(getattr(self, 'structure', -1))):
            self.bond_lists = True
        c.build_lists = 1 if getattr(self, 'bond_lists', False) else 0
        L.check(lib.eagcn_index_rows(C.byref(c), C.c_void_p(main.cuda_stream)), 'eagcn_index_rows')
        for t in (self.code, blob):
            t.record_stream(main)         # allocated on `side`, consumed on `main`
        self.c = c
        self._keep = (adj, rels, bonds)

    def ref(self):
        return C.byref(self.c)


# ------------------------------------------------------------------------------------------------
# layout conversion
# ------------------------------------------------------------------------------------------------
class _PackRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, index, layout, dense):
        dense = _need_cuda_f32(dense, 'afms')
        B, N, F = dense.shape
        if B != index.B or N != index.N or F != layout.width:
            raise L.EagcnHipError('activations %s do not match batch index [%d,%d,%d]'
                                  % (tuple(dense.shape), index.B, index.N, layout.width))
        out = torch.empty((index.T, layout.ld), dtype=torch.float32, device=dense.device)
        L.check(L.load().eagcn_pack_rows(index.ref(), _ptr(dense), F, C.byref(layout.c), _ptr(out), _stream()),
                'eagcn_pack_rows')
        ctx.index, ctx.layout, ctx.F = index, layout, F
        return out

    @staticmethod
    def backward(ctx, g):
        index, layout = ctx.index, ctx.layout
        g = g.contiguous()
        dense = torch.empty((index.B, index.N, ctx.F), dtype=torch.float32, device=g.device)
        L.check(L.load().eagcn_unpack_rows(index.ref(), _ptr(g), C.byref(layout.c), C.c_void_p(0), _ptr(dense),
                                           ctx.F, _stream()), 'eagcn_unpack_rows')
        return None, None, dense


class _UnpackRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, index, layout, packed, pad_row):
        F = layout.width
        dense = torch.empty((index.B, index.N, F), dtype=torch.float32, device=packed.device)
        L.check(L.load().eagcn_unpack_rows(index.ref(), _ptr(packed), C.byref(layout.c), _ptr(pad_row),
                                           _ptr(dense), F, _stream()), 'eagcn_unpack_rows')
        ctx.index, ctx.layout = index, layout
        ctx.has_pad = pad_row is not None
        return dense

    @staticmethod
    def backward(ctx, g):
        index, layout = ctx.index, ctx.layout
        g = g.contiguous()
        out = torch.empty((index.T, layout.ld), dtype=torch.float32, device=g.device)
        L.check(L.load().eagcn_pack_rows(index.ref(), _ptr(g), layout.width, C.byref(layout.c), _ptr(out),
                                         _stream()), 'eagcn_pack_rows')
        dpad = None
        if ctx.has_pad and ctx.needs_input_grad[3]:
            # gradient of the shared non-stored row = sum of the dense gradient over those rows
            nat = index.nat.to(torch.int64)
            ar = torch.arange(index.N, device=g.device).view(1, -1)
            mask = (ar >= nat.view(-1, 1)).to(g.dtype).unsqueeze(-1)
            dsum = (g * mask).sum(dim=(0, 1))
            dpad = torch.zeros(layout.ld, dtype=g.dtype, device=g.device)
            o_e = o_p = 0
            for w, p in zip(layout.widths, layout.pads):
                dpad[o_p:o_p + w] = dsum[o_e:o_e + w]
                o_e += w
                o_p += p
        return None, None, out, dpad


def pack_rows(index, layout, dense):
    return _PackRows.apply(index, layout, dense)


def unpack_rows(index, layout, packed, pad_row=None):
    return _UnpackRows.apply(index, layout, packed, pad_row)


# ------------------------------------------------------------------------------------------------
# graph-conv layer
# ------------------------------------------------------------------------------------------------
def _seed_fields(seed):
    """(host value, device pointer) of a dropout seed: an int, or a 1-element int64 DEVICE tensor -- the kernels then read the
    seed from device memory, so a captured launch draws fresh masks on every replay (graph mode)."""
    if isinstance(seed, torch.Tensor):
        if not seed.is_cuda or seed.dtype != torch.int64 or seed.numel() != 1:
            raise L.EagcnHipError('a device seed must be a 1-element int64 device tensor')
        return 0, seed.data_ptr()
    return int(seed) & (2 ** 63 - 1), None


class LayerSpec:
    """Static description of one multi-view layer (what the C struct needs besides pointers)."""

    def __init__(self, structure, widths, in_layout, dropout, bn_eps=1e-5, bn_momentum=0.1):
        self.structure = {'Concate': L.STRUCT_CONCATE, 'Weighted_sum': L.STRUCT_WEIGHTED}[structure]
        self.structure_name = structure
        self.widths = [int(w) for w in widths]
        self.K = len(self.widths)
        self.in_layout = in_layout
        self.dropout = float(dropout)
        self.bn_eps, self.bn_momentum = float(bn_eps), float(bn_momentum)
        self.fp = sum(L.pad16(w) for w in self.widths)
        if self.structure == L.STRUCT_CONCATE:
            self.out_layout = ColLayout(self.widths, [L.pad16(w) for w in self.widths])
        else:
            self.out_layout = ColLayout([self.widths[0]], [L.pad16(self.widths[0])])

    def cparams(self, training, seed, views, ave_w):
        """views: list of K dicts of tensors (att_w, self_r, W, bias, gamma, beta, run_mean, run_var)."""
        p = L.LayerParams()
        p.K, p.structure, p.training = self.K, self.structure, int(bool(training))
        for k, w in enumerate(self.widths):
            p.width[k] = w
        p.inp = self.in_layout.c
        p.dropout, p.bn_eps, p.bn_momentum = self.dropout, self.bn_eps, self.bn_momentum
        p.seed, p.seed_dev = _seed_fields(seed)
        for k, v in enumerate(views):
            p.att_w[k], p.self_r[k], p.W[k] = v['att_w'].data_ptr(), v['self_r'].data_ptr(), v['W'].data_ptr()
            p.bias[k], p.gamma[k], p.beta[k] = v['bias'].data_ptr(), v['gamma'].data_ptr(), v['beta'].data_ptr()
            p.run_mean[k], p.run_var[k] = v['run_mean'].data_ptr(), v['run_var'].data_ptr()
        p.ave_w = ave_w.data_ptr() if ave_w is not None else None
        return p


_PARAM_KEYS = ('att_w', 'self_r', 'W', 'bias', 'gamma', 'beta')


class _LayerFn(torch.autograd.Function):
    """x_packed, parameters -> (xout_packed, pad_row).  One C call forward, one backward."""

    @staticmethod
    def forward(ctx, index, spec, training, seed, buffers, x, ave_w, *flat):
        lib = L.load()
        K = spec.K
        x = x.contiguous()
        if x.shape != (index.T, spec.in_layout.ld):
            raise L.EagcnHipError('packed input is %s, expected %s' % (tuple(x.shape), (index.T, spec.in_layout.ld)))
        views = []
        for k in range(K):
            v = {name: flat[k * 6 + i] for i, name in enumerate(_PARAM_KEYS)}
            for name, t in v.items():
                if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                    raise L.EagcnHipError('parameter %s of view %d must be a contiguous fp32 device tensor' % (name, k))
            v['run_mean'], v['run_var'] = buffers[k]
            fin = spec.in_layout.width
            if tuple(v['W'].shape) != (fin, spec.widths[k]):
                raise L.EagcnHipError('view %d weight is %s, expected %s' % (k, tuple(v['W'].shape), (fin, spec.widths[k])))
            if v['att_w'].numel() != index.channels[k]:
                raise L.EagcnHipError('view %d attention weight has %d channels, relation tensor has %d'
                                      % (k, v['att_w'].numel(), index.channels[k]))
            views.append(v)
        p = spec.cparams(training, seed, views, ave_w)
        dev, T, fp = x.device, index.T, spec.fp
        f32 = dict(dtype=torch.float32, device=dev)
        P = torch.empty((T, fp), **f32)
        Y = torch.empty((T, fp), **f32)
        rscale = torch.empty((K, T), **f32)
        bn = torch.empty((4, fp), **f32)
        xout = torch.empty((T, spec.out_layout.ld), **f32)
        pad_row = torch.empty(spec.out_layout.ld, **f32)
        nbytes = lib.eagcn_layer_fwd_scratch_bytes(index.ref(), C.byref(p))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        w = L.LayerBufs()
        w.x, w.P, w.Y, w.rscale, w.bn = x.data_ptr(), P.data_ptr(), Y.data_ptr(), rscale.data_ptr(), bn.data_ptr()
        w.xout, w.pad_row, w.scratch, w.scratch_bytes = xout.data_ptr(), pad_row.data_ptr(), scratch.data_ptr(), nbytes
        L.check(lib.eagcn_layer_forward(index.ref(), C.byref(p), C.byref(w), _stream()), 'eagcn_layer_forward')
        ctx.index, ctx.spec, ctx.training, ctx.seed, ctx.buffers = index, spec, training, seed, buffers
        ctx.save_for_backward(x, P, Y, rscale, bn, ave_w, *flat)
        if spec.structure == L.STRUCT_CONCATE:
            ctx.mark_non_differentiable(pad_row)
        return xout, pad_row

    @staticmethod
    def backward(ctx, dxout, dpad):
        lib = L.load()
        index, spec = ctx.index, ctx.spec
        x, P, Y, rscale, bn, ave_w, *flat = ctx.saved_tensors
        K = spec.K
        views = []
        for k in range(K):
            v = {name: flat[k * 6 + i] for i, name in enumerate(_PARAM_KEYS)}
            v['run_mean'], v['run_var'] = ctx.buffers[k]
            views.append(v)
        p = spec.cparams(ctx.training, ctx.seed, views, ave_w)
        dev = x.device
        dxout = dxout.contiguous()
        grads = [torch.empty_like(t) for t in flat]
        dave = torch.empty_like(ave_w) if ave_w is not None else None
        g = L.LayerGrads()
        for k in range(K):
            gk = grads[k * 6:(k + 1) * 6]
            g.datt_w[k], g.dself_r[k], g.dW[k] = gk[0].data_ptr(), gk[1].data_ptr(), gk[2].data_ptr()
            g.dbias[k], g.dgamma[k], g.dbeta[k] = gk[3].data_ptr(), gk[4].data_ptr(), gk[5].data_ptr()
        g.dave_w = dave.data_ptr() if dave is not None else None
        need_dx = ctx.needs_input_grad[5]
        dx = torch.empty_like(x) if need_dx else None
        nbytes = lib.eagcn_layer_bwd_scratch_bytes(index.ref(), C.byref(p))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        w = L.LayerBufs()
        w.x, w.P, w.Y, w.rscale, w.bn = x.data_ptr(), P.data_ptr(), Y.data_ptr(), rscale.data_ptr(), bn.data_ptr()
        w.scratch, w.scratch_bytes = scratch.data_ptr(), nbytes
        dpad_ptr = C.c_void_p(0)
        if spec.structure == L.STRUCT_WEIGHTED and dpad is not None:
            dpad = dpad.contiguous()
            dpad_ptr = _ptr(dpad)
        L.check(lib.eagcn_layer_backward(index.ref(), C.byref(p), C.byref(w), _ptr(dxout), dpad_ptr, _ptr(dx),
                                         C.byref(g), _stream()), 'eagcn_layer_backward')
        return (None, None, None, None, None, dx, dave, *grads)


def layer_forward(index, spec, training, seed, buffers, x, ave_w, flat_params):
    return _LayerFn.apply(index, spec, training, seed, buffers, x, ave_w, *flat_params)


class _GatFn(torch.autograd.Function):
    """The GAT baseline layer (reference layers.py:99-203) on packed rows: (x, W, a) -> xout.  One C call each way."""

    @staticmethod
    def forward(ctx, index, fin, ld_in, F, training, seed, dropout, att_dropout, alpha, x, W, a):
        lib = L.load()
        if not getattr(index, 'bond_lists', False):
            raise L.EagcnHipError('the GAT layer needs a batch index with bond lists: BatchIndex(..., bond_lists=True)')
        x, W, a = x.contiguous(), W.contiguous(), a.contiguous()
        if x.shape != (index.T, ld_in) or tuple(W.shape) != (fin, F) or a.numel() != 2 * F:
            raise L.EagcnHipError('GAT layer: x %s W %s a %s for fin=%d ld=%d F=%d' % (tuple(x.shape), tuple(W.shape), tuple(a.shape), fin, ld_in, F))
        p = L.GatParams()
        p.fin, p.ld_in, p.F, p.training = fin, ld_in, F, int(bool(training))
        p.alpha, p.att_dropout, p.dropout = float(alpha), float(att_dropout), float(dropout)
        p.seed, p.seed_dev = _seed_fields(seed)
        p.W, p.a = W.data_ptr(), a.data_ptr()
        Fp = (F + 15) // 16 * 16
        f32 = dict(dtype=torch.float32, device=x.device)
        h = torch.empty((index.T, Fp), **f32)
        s12 = torch.empty((2, index.T), **f32)
        xout = torch.empty((index.T, Fp), **f32)
        L.check(lib.eagcn_gat_forward(index.ref(), C.byref(p), x.data_ptr(), h.data_ptr(), s12.data_ptr(), xout.data_ptr(),
                                      _stream()), 'eagcn_gat_forward')
        ctx.index, ctx.p_args = index, (fin, ld_in, F, training, seed, dropout, att_dropout, alpha)
        ctx.save_for_backward(x, W, a, h, s12, xout)
        return xout

    @staticmethod
    def backward(ctx, dxout):
        lib = L.load()
        x, W, a, h, s12, xout = ctx.saved_tensors
        fin, ld_in, F, training, seed, dropout, att_dropout, alpha = ctx.p_args
        index = ctx.index
        p = L.GatParams()
        p.fin, p.ld_in, p.F, p.training = fin, ld_in, F, int(bool(training))
        p.alpha, p.att_dropout, p.dropout = float(alpha), float(att_dropout), float(dropout)
        p.seed, p.seed_dev = _seed_fields(seed)
        p.W, p.a = W.data_ptr(), a.data_ptr()
        dxout = dxout.contiguous()
        dW, da = torch.empty_like(W), torch.empty_like(a)
        dx = torch.empty_like(x) if ctx.needs_input_grad[9] else None
        nbytes = lib.eagcn_gat_scratch_bytes(index.ref(), F)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        L.check(lib.eagcn_gat_backward(index.ref(), C.byref(p), x.data_ptr(), h.data_ptr(), s12.data_ptr(), xout.data_ptr(),
                                       dxout.data_ptr(), _ptr(dx) if dx is not None else C.c_void_p(0), dW.data_ptr(),
                                       da.data_ptr(), scratch.data_ptr(), nbytes, _stream()), 'eagcn_gat_backward')
        return (None,) * 9 + (dx, dW, da)


def gat_layer(index, fin, ld_in, F, training, seed, dropout, att_dropout, alpha, x, W, a):
    return _GatFn.apply(index, fin, ld_in, F, training, seed, dropout, att_dropout, alpha, x, W, a)


def attention_dense(index, att_weights):
    """A_weight of layers.py:318: [K,B,N,N] stack of sigmoid(w_k[type]) * adj (no gradient)."""
    lib = L.load()
    p = L.LayerParams()
    p.K = index.K
    for k, a in enumerate(att_weights):
        p.att_w[k] = a.data_ptr()
    out = torch.empty((index.K, index.B, index.N, index.N), dtype=torch.float32, device=index.device)
    L.check(lib.eagcn_attention_dense(index.ref(), C.byref(p), _ptr(out), _stream()), 'eagcn_attention_dense')
    return out


# ------------------------------------------------------------------------------------------------
# read-out
# ------------------------------------------------------------------------------------------------
class _Readout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, index, layout, mode, size, x, pad_row):
        F = layout.width
        g = torch.empty((index.B, F), dtype=torch.float32, device=x.device)
        if mode == 1:
            size = size.to(device=x.device, dtype=torch.int64).contiguous()
        L.check(L.load().eagcn_readout_forward(index.ref(), _ptr(x), C.byref(layout.c), _ptr(pad_row),
                                               _ptr(size) if mode == 1 else C.c_void_p(0), mode, _ptr(g), F,
                                               _stream()), 'eagcn_readout_forward')
        ctx.index, ctx.layout, ctx.mode, ctx.size = index, layout, mode, size
        ctx.has_pad = pad_row is not None
        return g

    @staticmethod
    def backward(ctx, dg):
        index, layout = ctx.index, ctx.layout
        dg = dg.contiguous()
        dx = torch.empty((index.T, layout.ld), dtype=torch.float32, device=dg.device)
        dpad = None
        if ctx.has_pad and ctx.needs_input_grad[5]:
            dpad = torch.empty(layout.ld, dtype=torch.float32, device=dg.device)
        L.check(L.load().eagcn_readout_backward(index.ref(), _ptr(dg), C.byref(layout.c),
                                                _ptr(ctx.size) if ctx.mode == 1 else C.c_void_p(0), ctx.mode,
                                                layout.width, _ptr(dx), _ptr(dpad), _stream()),
                'eagcn_readout_backward')
        return None, None, None, None, dx, dpad


# ------------------------------------------------------------------------------------------------
# Diff_Pooling read-out (molfp_mode='pool'): csrc/pool.hip + the flat fp32 GEMM
# ------------------------------------------------------------------------------------------------
POOL_MODES = {'attention': 0, 'gcn': 1, 'gat': 2}


class _PoolAttention(torch.autograd.Function):
    """The attention matrix the last layer returns (layers.py:319-324 / 250-253 / 189) as packed rows [T][lda], with its
    backward into ave_A.weight, the layer's self_r and the views' att.weight."""

    @staticmethod
    def forward(ctx, index, mode, ave_a, self_r, *att_w):
        T, lda = max(index.T, 1), L.pad4(index.N)
        dev = index.device
        A = torch.empty((T, lda), dtype=torch.float32, device=dev)
        rinv = torch.empty(T, dtype=torch.float32, device=dev)
        padsum = torch.empty(T, dtype=torch.float32, device=dev)
        p = L.PoolAtt()
        p.mode = mode
        p.K = len(att_w) if mode == 0 else index.K
        if mode == 0:
            ave_a, self_r = _need_cuda_f32(ave_a, 'ave_A.weight').contiguous(), _need_cuda_f32(self_r, 'self_r').contiguous()
            att_w = [_need_cuda_f32(a, 'att.weight').contiguous() for a in att_w]
            p.ave_a, p.self_r = ave_a.data_ptr(), self_r.data_ptr()
            for k, a in enumerate(att_w):
                p.att_w[k], p.att_c[k] = a.data_ptr(), a.numel()
        L.check(L.load().eagcn_pool_attention_forward(index.ref(), C.byref(p), _ptr(A), lda, _ptr(rinv), _ptr(padsum),
                                                      _stream()), 'eagcn_pool_attention_forward')
        ctx.index, ctx.mode = index, mode
        if mode == 0:
            ctx.save_for_backward(A, rinv, ave_a, self_r, *att_w)
        ctx.mark_non_differentiable(rinv, padsum)
        return A, rinv, padsum

    @staticmethod
    def backward(ctx, dA, _drinv, _dpadsum):
        if ctx.mode != 0:
            return (None,) * 4
        A, rinv, ave_a, self_r, *att_w = ctx.saved_tensors
        index = ctx.index
        lib = L.load()
        p = L.PoolAtt()
        p.mode, p.K = 0, len(att_w)
        p.ave_a, p.self_r = ave_a.data_ptr(), self_r.data_ptr()
        dave, dself = torch.empty_like(ave_a), torch.empty_like(self_r)
        datt = [torch.empty_like(a) for a in att_w]
        p.dave_a, p.dself_r = dave.data_ptr(), dself.data_ptr()
        for k, a in enumerate(att_w):
            p.att_w[k], p.att_c[k], p.datt_w[k] = a.data_ptr(), a.numel(), datt[k].data_ptr()
        nbytes = lib.eagcn_pool_scratch_bytes()
        scratch = _scratch(index.device, nbytes)
        L.check(lib.eagcn_pool_attention_backward(index.ref(), C.byref(p), _ptr(A), A.shape[1], _ptr(rinv),
                                                  _ptr(dA.contiguous()), scratch.data_ptr(), nbytes, _stream()),
                'eagcn_pool_attention_backward')
        return (None, None, dave, dself, *datt)


class _PoolMix(torch.autograd.Function):
    """AX = A . x over the stored rows (layers.py:39): [T][F] exact columns."""

    @staticmethod
    def forward(ctx, index, layout, A, padsum, x, pad_row):
        F = layout.width
        # (zeros: with a capacity-sized index the rows beyond the batch's atoms are not written, and the flat GEMMs behind
        #  this matrix run over all of them)
        AX = torch.zeros((max(index.T, 1), F), dtype=torch.float32, device=x.device)
        L.check(L.load().eagcn_pool_mix_forward(index.ref(), C.byref(layout.c), _ptr(A), A.shape[1], _ptr(padsum), _ptr(x),
                                                _ptr(pad_row), _ptr(AX), F, _stream()), 'eagcn_pool_mix_forward')
        ctx.index, ctx.layout = index, layout
        ctx.has_pad = pad_row is not None
        ctx.save_for_backward(A, padsum, x, pad_row)
        return AX

    @staticmethod
    def backward(ctx, dAX):
        A, padsum, x, pad_row = ctx.saved_tensors
        index, layout = ctx.index, ctx.layout
        dAX = dAX.contiguous()
        dA = torch.empty_like(A) if ctx.needs_input_grad[2] else None
        dx = torch.zeros_like(x) if ctx.needs_input_grad[4] else None
        dpad = torch.empty_like(pad_row) if ctx.has_pad and ctx.needs_input_grad[5] else None
        L.check(L.load().eagcn_pool_mix_backward(index.ref(), C.byref(layout.c), _ptr(A), A.shape[1], _ptr(padsum), _ptr(x),
                                                 _ptr(pad_row), _ptr(dAX), layout.width, _ptr(dA), _ptr(dx), _ptr(dpad),
                                                 _stream()), 'eagcn_pool_mix_backward')
        return None, None, dA, None, dx, dpad


class _PoolReduce(torch.autograd.Function):
    """Z [T][F+P] = (A.x).[Wf | Ws]  ->  g [B][F] = sum_p relu(S^T relu(Z[:, :F])), S = softmax(Z[:, F:])
    (layers.py:499-503 and the sum over clusters of models.py:106)."""

    @staticmethod
    def forward(ctx, index, F, P, Z):
        dev = Z.device
        S = torch.empty((max(index.T, 1), P), dtype=torch.float32, device=dev)
        Pm = torch.empty((index.B, P, F), dtype=torch.float32, device=dev)
        g = torch.empty((index.B, F), dtype=torch.float32, device=dev)
        L.check(L.load().eagcn_pool_reduce_forward(index.ref(), _ptr(Z), Z.shape[1], F, P, _ptr(S), _ptr(Pm), _ptr(g),
                                                   _stream()), 'eagcn_pool_reduce_forward')
        ctx.index, ctx.F, ctx.P = index, F, P
        ctx.save_for_backward(Z, S, Pm)
        return g

    @staticmethod
    def backward(ctx, dg):
        Z, S, Pm = ctx.saved_tensors
        dZ = torch.zeros_like(Z)                            # (rows beyond the batch's atoms stay zero: see _PoolMix.forward)
        L.check(L.load().eagcn_pool_reduce_backward(ctx.index.ref(), _ptr(Z), Z.shape[1], ctx.F, ctx.P, _ptr(S), _ptr(Pm),
                                                    _ptr(dg.contiguous()), _ptr(dZ), _stream()), 'eagcn_pool_reduce_backward')
        return None, None, None, dZ


def pool_readout(index, layout, x, pad_row, w_feature, w_assign, mode='attention', att_w=(), ave_a=None, self_r=None):
    """models.py:104-106 for molfp_mode='pool': sum over the clusters of Diff_Pooling's pooled features.  `mode` names the
    family of the last layer (its attention matrix is rebuilt from the batch index and the layer's parameters)."""
    if layout.width != w_feature.shape[0] or w_assign.shape[0] != w_feature.shape[0]:
        raise L.EagcnHipError('pool_readout: %d feature columns, weights [%d,..] / [%d,..]'
                              % (layout.width, w_feature.shape[0], w_assign.shape[0]))
    F, P = int(w_feature.shape[1]), int(w_assign.shape[1])
    if not 1 <= P <= L.POOL_MAX:
        raise L.EagcnHipError('pool_readout: %d clusters (1..%d)' % (P, L.POOL_MAX))
    if index.T == 0:                                   # no atom in the whole batch: every pooled feature is relu(0)
        return torch.zeros((index.B, F), dtype=torch.float32, device=x.device)
    A, _rinv, padsum = _PoolAttention.apply(index, POOL_MODES[mode], ave_a, self_r, *att_w)
    AX = _PoolMix.apply(index, layout, A, padsum, x, pad_row)
    Z = dense_mm(AX, torch.cat([w_feature, w_assign], dim=1), index)   # one flat GEMM for both bases (layers.py:40)
    return _PoolReduce.apply(index, F, P, Z)


def readout(index, layout, x, pad_row, mode='sum', size=None):
    return _Readout.apply(index, layout, 1 if mode == 'ave' else 0, size, x, pad_row)


class _DenseMM(torch.autograd.Function):
    """y = x @ W on the fp32 matrix cores (reference Dense.forward, layers.py:382-387) with its
    two backward products (dx = dy @ W^T, dW = x^T @ dy)."""

    @staticmethod
    def forward(ctx, x, w, index=None):
        x = _need_cuda_f32(x, 'dense input')
        w = _need_cuda_f32(w, 'dense weight')
        ctx.save_for_backward(x, w)
        ctx.index = index
        return gemm(x, w, rows_of=index, which=0)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = gemm(dy, w, tb=True, rows_of=ctx.index, which=0) if ctx.needs_input_grad[0] else None
        dw = gemm(x, dy, ta=True, rows_of=ctx.index, which=2) if ctx.needs_input_grad[1] else None
        return dx, dw, None


def dense_mm(x, w, index=None):
    """x @ w; with `index` the rows of x are the packed rows of that batch index: their count is read on the device (a
    capacity-sized index of graph mode holds far fewer atoms than rows), rows beyond it are neither read nor written."""
    return _DenseMM.apply(x, w, index)


def _pad2(t, rows, cols):
    """Zero-pad a matrix so that the flagged dimensions are multiples of 4 (the GEMM loads
    contiguous runs as float4 and walks K in steps of 4); zero padding leaves the product unchanged."""
    pr = (-t.shape[0]) % 4 if rows else 0
    pc = (-t.shape[1]) % 4 if cols else 0
    return t if (pr == 0 and pc == 0) else torch.nn.functional.pad(t, (0, pc, 0, pr))


def gemm(a, b, ta=False, tb=False, rows_of=None, which=0):
    """C = op(A) @ op(B) on the fp32 matrix cores (no autograd; used by the head and by tests).
    ta: A is stored [K,M]; tb: B is stored [N,K].  rows_of (a batch index): the extent `which` (0: rows of A and C, 2: the
    reduction length) is the index's packed row count, read on the device."""
    M = a.shape[1] if ta else a.shape[0]
    Ka = a.shape[0] if ta else a.shape[1]
    N = b.shape[0] if tb else b.shape[1]
    Kb = b.shape[1] if tb else b.shape[0]
    if Ka != Kb:
        raise L.EagcnHipError('gemm: inner dimensions differ (%d vs %d)' % (Ka, Kb))
    a = _pad2(a.contiguous(), rows=ta, cols=True)         # [M,K]: pad K ; [K,M]: pad K and M
    b = _pad2(b.contiguous(), rows=not tb, cols=True)     # [K,N]: pad K and N ; [N,K]: pad K
    Mp = a.shape[1] if ta else M
    Np = N if tb else b.shape[1]
    Kp = a.shape[0] if ta else a.shape[1]
    if rows_of is not None:
        # (zeros: the rows beyond the device-side count are not written)
        c = torch.zeros((Mp, Np), dtype=torch.float32, device=a.device) if which == 0 else \
            torch.empty((Mp, Np), dtype=torch.float32, device=a.device)
        L.check(L.load().eagcn_gemm_f32_dev(int(ta), int(tb), Mp, Np, Kp, _ptr(a), a.shape[1], _ptr(b), b.shape[1], _ptr(c), Np,
                                            C.c_void_p(int(rows_of.c.meta) + 4 * L.META_T), int(which), _stream()),
                'eagcn_gemm_f32_dev')
        return c if (Mp == M and Np == N) else c[:M, :N].contiguous()
    c = torch.empty((Mp, Np), dtype=torch.float32, device=a.device)
    L.check(L.load().eagcn_gemm_f32(int(ta), int(tb), Mp, Np, Kp, _ptr(a), a.shape[1], _ptr(b), b.shape[1],
                                    _ptr(c), Np, _stream()), 'eagcn_gemm_f32')
    return c if (Mp == M and Np == N) else c[:M, :N].contiguous()


# ------------------------------------------------------------------------------------------------
# whole model: one C call forward, one backward
# ------------------------------------------------------------------------------------------------
def fill_head_params(hp, h, dropout):
    """eagcn_head_params (include/eagcn_hip.h) of the head modules h = {den1, den2, den3, Graph_BN, bn_den1, bn_den2}."""
    hp.f_in, hp.n_den1 = h['den1'].weight.shape
    hp.n_den2, hp.nclass = h['den3'].weight.shape
    hp.dropout = float(dropout)
    hp.bn_eps, hp.bn_momentum = float(h['Graph_BN'].eps), float(h['Graph_BN'].momentum)
    hp.den1_w, hp.den2_w, hp.den3_w = (h['den1'].weight.data_ptr(), h['den2'].weight.data_ptr(),
                                       h['den3'].weight.data_ptr())
    for pre, mod in (('gbn', h['Graph_BN']), ('bn1', h['bn_den1']), ('bn2', h['bn_den2'])):
        setattr(hp, pre + '_w', mod.weight.data_ptr())
        setattr(hp, pre + '_b', mod.bias.data_ptr())
        setattr(hp, pre + '_rm', mod.running_mean.data_ptr())
        setattr(hp, pre + '_rv', mod.running_var.data_ptr())
    return hp


_scratch_cache = {}


def _scratch(device, nbytes):
    """Transient workspace, reused across calls on the same (device, stream): all users are
    stream-ordered, so the previous call's kernels are done with it before the next ones start."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    t = _scratch_cache.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _scratch_cache[key] = t
        from .parallel import register_stats_buffer
        register_stats_buffer(t)                      # (pieces of it are handed to the sync-BatchNorm hook)
    return t


class ModelPlan:
    """Static description of an EAGCN module for the model-level entry points: parameter order,
    layer specs with their chained column layouts, gradient-buffer offsets."""

    def __init__(self, layers, head, n_afeat, molfp_mode, dropout, stats=None, fuse_readout=False):
        # layers: list of GraphConv_Layer modules; head: dict of modules; stats: parallel.StatsAllReducer (sync-BatchNorm) or None
        # fuse_readout: a Concate top layer's output matrix is only built on request (eagcn_model.fuse_readout)
        self.stats = stats
        self.fuse_readout = bool(fuse_readout)
        self.layers = layers
        self.head = head
        self.molfp = 1 if molfp_mode == 'ave' else 0
        self.specs = []
        layout = ColLayout.single(n_afeat)
        for layer in layers:
            spec = layer.spec_for(layout)
            self.specs.append(spec)
            layout = spec.out_layout
        self.last_layout = layout
        self.params = []          # flat, fixed order
        self.layer_slices = []    # per layer: (start index of view params, index of ave_w or None)
        for layer in layers:
            start = len(self.params)
            for blk in layer.blocks():
                self.params.extend(blk.hot_params())
            ave = None
            if layer.structure == 'Weighted_sum':
                ave = len(self.params)
                self.params.append(layer.ave.weight)
            self.layer_slices.append((start, ave))
        self.head_start = len(self.params)
        h = head
        self.params.extend([h['den1'].weight, h['den2'].weight, h['den3'].weight,
                            h['Graph_BN'].weight, h['Graph_BN'].bias, h['bn_den1'].weight, h['bn_den1'].bias,
                            h['bn_den2'].weight, h['bn_den2'].bias])
        self.sizes = [p.numel() for p in self.params]
        self.shapes = [tuple(p.shape) for p in self.params]
        # every parameter's gradient starts on a 16-byte boundary of the flat buffer (a 1-element self_r in front of a weight
        # matrix would otherwise leave the matrix 4-byte aligned: the GEMM epilogue that writes blockK.graph_conv.weight.grad
        # uses 16-byte stores).  The few padding words stay zero.
        self.offsets = [0]
        for n in self.sizes:
            self.offsets.append(self.offsets[-1] + L.pad4(n))
        self.trigger = None
        self.flat_grad = None
        self._cm_cache = {}
        self._ptr_tensors = list(self.params)
        for layer in layers:
            for blk in layer.blocks():
                self._ptr_tensors += [blk.batch_norm.bn.running_mean, blk.batch_norm.bn.running_var]
        for n in ('Graph_BN', 'bn_den1', 'bn_den2'):
            self._ptr_tensors += [head[n].running_mean, head[n].running_var]
        self.nbt = [blk.batch_norm.bn.num_batches_tracked for layer in layers for blk in layer.blocks()] + \
                   [h['Graph_BN'].num_batches_tracked, h['bn_den1'].num_batches_tracked,
                    h['bn_den2'].num_batches_tracked]
        self.nbt_pending = 0          # graph mode counts batches on the host (see flush_nbt)

    def grad_views(self, flat):
        """Per-parameter views (parameter shapes) of a flat gradient buffer laid out by `offsets`."""
        return [flat[o:o + n] if len(sh) == 1 else flat[o:o + n].view(sh)
                for o, n, sh in zip(self.offsets, self.sizes, self.shapes)]

    def flush_nbt(self):
        """Write the batches counted on the host by graph mode into the num_batches_tracked buffers (they are not
        read by the computation: momentum is a number, reference layers.py:403).  Called by EAGCN.state_dict()."""
        if self.nbt_pending:
            torch._foreach_add_(self.nbt, self.nbt_pending)
            self.nbt_pending = 0

    def cmodel(self, training, seed, dropout):
        """The C description of the model for this call.  The struct is cached per (training, dropout)
        and rebuilt only when a parameter / buffer pointer changed (e.g. after .to() or a re-assigned
        .data); per call only the dropout seeds are refreshed."""
        key = (bool(training), float(dropout), tuple(float(l.dropout) for l in self.layers))
        ptrs = [t.data_ptr() for t in self._ptr_tensors]
        hit = self._cm_cache.get(key)
        if hit is None or hit[1] != ptrs:
            hit = (self._build_cmodel(training, dropout), ptrs)
            self._cm_cache[key] = hit
        m = hit[0]
        seed = int(seed)
        m.head_seed = (seed + 0x51ED27) & (2 ** 63 - 1)
        for l in range(len(self.layers)):
            m.layer[l].seed = (seed + 7919 * (l + 1)) & (2 ** 63 - 1)
        return m

    def _build_cmodel(self, training, dropout):
        seed = 0
        m = L.Model()
        m.n_layers, m.molfp_mode, m.training = len(self.layers), self.molfp, int(bool(training))
        m.head_seed = (int(seed) + 0x51ED27) & (2 ** 63 - 1)
        for l, (layer, spec) in enumerate(zip(self.layers, self.specs)):
            spec.dropout = float(layer.dropout)
            views = []
            for blk in layer.blocks():
                bn = blk.batch_norm.bn
                views.append({'att_w': blk.att.weight, 'self_r': blk.self_r, 'W': blk.graph_conv.weight,
                              'bias': blk.graph_conv.bias, 'gamma': bn.weight, 'beta': bn.bias,
                              'run_mean': bn.running_mean, 'run_var': bn.running_var})
            ave = layer.ave.weight if layer.structure == 'Weighted_sum' else None
            m.layer[l] = spec.cparams(training, (int(seed) + 7919 * (l + 1)) & (2 ** 63 - 1), views, ave)
        fill_head_params(m.head, self.head, dropout)
        m.fuse_readout = int(self.fuse_readout)
        if self.stats is not None:                    # sync-BatchNorm: cross-rank sums through eagcn_amd.parallel.StatsAllReducer
            m.stats_hook = C.cast(self.stats.cfn, C.c_void_p)
            m.stats_world = int(self.stats.world())
        return m


class _ModelFn(torch.autograd.Function):
    """(afm, every hot parameter) -> (out, graph_representation): one call into
    eagcn_model_forward, one into eagcn_model_backward."""

    @staticmethod
    def forward(ctx, plan, index, holder, training, seed, dropout, size, afm, trigger, *params):
        lib = L.load()
        if ctx.needs_input_grad[7]:
            raise L.EagcnHipError('the model-level engine does not produce d/d(afms) (the reference training loop never '
                                  'asks for it); use EAGCN.forward_composed, whose layer-level ops return it')
        afm = _need_cuda_f32(afm, 'afms')
        ctx.direct = trigger is not None
        if ctx.direct:
            params = plan.params
        for i, t in enumerate(params):
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise L.EagcnHipError('parameter %d must be a contiguous fp32 device tensor' % i)
        if afm.shape != (index.B, index.N, plan.specs[0].in_layout.width):
            raise L.EagcnHipError('afms is %s, expected %s' % (tuple(afm.shape), (index.B, index.N, plan.specs[0].in_layout.width)))
        m = plan.cmodel(training, seed, dropout)
        dev = afm.device
        if plan.molfp:
            size = size.to(device=dev, dtype=torch.int64).contiguous()
        sbytes = lib.eagcn_model_saved_bytes(index.ref(), C.byref(m))
        wbytes = lib.eagcn_model_scratch_bytes(index.ref(), C.byref(m))
        saved = torch.empty(sbytes, dtype=torch.uint8, device=dev)
        scratch = _scratch(dev, wbytes)
        out = torch.empty((index.B, m.head.nclass), dtype=torch.float32, device=dev)
        graph_rep = torch.empty((index.B, m.head.n_den2), dtype=torch.float32, device=dev)
        L.check(lib.eagcn_model_forward(index.ref(), C.byref(m), _ptr(afm), _ptr(size) if plan.molfp else C.c_void_p(0),
                                        _ptr(saved), sbytes, _ptr(scratch), scratch.numel(), _ptr(out),
                                        _ptr(graph_rep), _stream()), 'eagcn_model_forward')
        if holder is not None:
            xo, po, ld = C.c_size_t(), C.c_size_t(), C.c_int()
            lib.eagcn_model_atom_rep(index.ref(), C.byref(m), C.byref(xo), C.byref(po), C.byref(ld))
            T = index.T
            holder['xout'] = saved[xo.value:xo.value + 4 * T * ld.value].view(torch.float32).view(T, ld.value)
            holder['pad_row'] = saved[po.value:po.value + 4 * ld.value].view(torch.float32)
            if m.fuse_readout:
                # the forward did not build the top layer's output matrix (its relu / dropout / mask ran inside the read-out):
                # it is built from the saved pre-BatchNorm matrix the first time the atom representations are touched
                mcopy = L.Model.from_buffer_copy(m)

                def materialize(index=index, mcopy=mcopy, saved=saved, sbytes=sbytes):
                    L.check(lib.eagcn_model_atom_rep_materialize(index.ref(), C.byref(mcopy), _ptr(saved), sbytes, _stream()),
                            'eagcn_model_atom_rep_materialize')
                holder['materialize'] = materialize
        # private copy: the cached struct is re-seeded by the next forward call
        ctx.plan, ctx.index, ctx.cmodel, ctx.saved_blob, ctx.size = plan, index, L.Model.from_buffer_copy(m), saved, size
        if not ctx.direct:
            ctx.save_for_backward(*params)
        return out, graph_rep

    @staticmethod
    def backward(ctx, dout, dgraph_rep):
        lib = L.load()
        plan, index, m, saved = ctx.plan, ctx.index, ctx.cmodel, ctx.saved_blob
        if not ctx.direct:
            ctx.saved_tensors                          # version check of the parameters by autograd
        dev = saved.device
        dout = dout.contiguous()
        dgr = dgraph_rep.contiguous() if dgraph_rep is not None else None
        flat = torch.zeros(plan.offsets[-1], dtype=torch.float32, device=dev)
        base = flat.data_ptr()

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
        hs = plan.head_start
        for j, name in enumerate(('d_den1_w', 'd_den2_w', 'd_den3_w', 'd_gbn_w', 'd_gbn_b', 'd_bn1_w', 'd_bn1_b',
                                  'd_bn2_w', 'd_bn2_b')):
            setattr(hg, name, gptr(hs + j))
        scratch = _scratch(dev, lib.eagcn_model_scratch_bytes(index.ref(), C.byref(m)))
        L.check(lib.eagcn_model_backward(index.ref(), C.byref(m), _ptr(ctx.size) if plan.molfp else C.c_void_p(0),
                                         _ptr(saved), saved.numel(), _ptr(scratch), scratch.numel(), _ptr(dout),
                                         _ptr(dgr), lg, C.byref(hg), _stream()), 'eagcn_model_backward')
        ctx.saved_blob = None
        grads = plan.grad_views(flat)
        if ctx.direct:
            # gradients are delivered straight into .grad as views of ONE flat buffer (accumulating if a
            # gradient is already there), bypassing ~70 AccumulateGrad nodes
            for p, g in zip(plan.params, grads):
                if not p.requires_grad:          # frozen buffers riding in plan.params (Vanilla_GCN's constant weights)
                    continue
                if p.grad is None:
                    p.grad = g
                else:
                    p.grad.add_(g)
            plan.flat_grad = flat
            return (None,) * 10
        return (None, None, None, None, None, None, None, None, None, *grads)


def model_forward(plan, index, holder, training, seed, dropout, size, afm, direct=False):
    if direct and torch.is_grad_enabled():
        if plan.trigger is None or plan.trigger.device != afm.device:
            plan.trigger = torch.zeros((), dtype=torch.float32, device=afm.device, requires_grad=True)
        return _ModelFn.apply(plan, index, holder, training, seed, dropout, size, afm, plan.trigger)
    return _ModelFn.apply(plan, index, holder, training, seed, dropout, size, afm, None, *plan.params)


# ------------------------------------------------------------------------------------------------
# the head alone (reference models.py:112-120) behind fingerprints formed by layer-level ops: GAT baseline, Diff_Pooling read-out
# ------------------------------------------------------------------------------------------------
HEAD_PARAMS = (('den1', 'weight'), ('den2', 'weight'), ('den3', 'weight'), ('Graph_BN', 'weight'), ('Graph_BN', 'bias'),
               ('bn_den1', 'weight'), ('bn_den1', 'bias'), ('bn_den2', 'weight'), ('bn_den2', 'bias'))


class _HeadFn(torch.autograd.Function):
    """(g [B, f_in], the nine head parameters) -> (out, graph_representation): one call into eagcn_head_forward (csrc/head.hip:
    Graph_BN statistics + three fused BatchNorm / product stages of csrc/head2.hip), one into eagcn_head_backward; running
    statistics are updated in place by the kernels."""

    @staticmethod
    def forward(ctx, mods, training, seed, dropout, g, *params):
        lib = L.load()
        g = _need_cuda_f32(g, 'molecule fingerprints')
        for (mn, pn), t in zip(HEAD_PARAMS, params):
            if t is not getattr(mods[mn], pn):
                raise L.EagcnHipError('head: parameter %s.%s is not the module\'s own tensor' % (mn, pn))
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise L.EagcnHipError('head: %s.%s must be a contiguous fp32 device tensor' % (mn, pn))
        for mn in ('den1', 'den2', 'den3'):
            if getattr(mods[mn], 'bias', None) is not None:
                raise L.EagcnHipError('head: Dense layers with a bias are not supported (models.py:86-88 builds them without)')
        hp = fill_head_params(L.HeadParams(), mods, dropout)
        B = int(g.shape[0])
        if g.shape != (B, hp.f_in):
            raise L.EagcnHipError('head: fingerprints are %s, den1 expects [B, %d]' % (tuple(g.shape), hp.f_in))
        host_seed, seed_dev = _seed_fields(seed)
        dev = g.device
        sbytes, wbytes = lib.eagcn_head_saved_bytes(C.byref(hp), B), lib.eagcn_head_scratch_bytes(C.byref(hp), B)
        saved = torch.empty(sbytes, dtype=torch.uint8, device=dev)
        scratch = _scratch(dev, wbytes)
        out = torch.empty((B, hp.nclass), dtype=torch.float32, device=dev)
        graph_rep = torch.empty((B, hp.n_den2), dtype=torch.float32, device=dev)
        L.check(lib.eagcn_head_forward(C.byref(hp), B, int(bool(training)), host_seed, C.c_void_p(seed_dev or 0), _ptr(g),
                                       _ptr(saved), sbytes, _ptr(scratch), scratch.numel(), _ptr(out), _ptr(graph_rep),
                                       _stream()), 'eagcn_head_forward')
        if training:
            torch._foreach_add_([mods[n].num_batches_tracked for n in ('Graph_BN', 'bn_den1', 'bn_den2')], 1)
        ctx.hp, ctx.B, ctx.training, ctx.seed, ctx.seed_keep = hp, B, int(bool(training)), (host_seed, seed_dev), seed
        ctx.saved_blob = saved
        ctx.save_for_backward(g, *params)
        return out, graph_rep

    @staticmethod
    def backward(ctx, dout, dgraph_rep):
        lib = L.load()
        g, *params = ctx.saved_tensors                 # (version check of the parameters by autograd)
        hp, B, saved = ctx.hp, ctx.B, ctx.saved_blob
        dev = g.device
        dout = dout.contiguous()
        dgr = dgraph_rep.contiguous() if dgraph_rep is not None else None
        grads = [torch.empty_like(p) for p in params]
        hg = L.HeadGrads()
        for name, t in zip(('d_den1_w', 'd_den2_w', 'd_den3_w', 'd_gbn_w', 'd_gbn_b', 'd_bn1_w', 'd_bn1_b', 'd_bn2_w', 'd_bn2_b'),
                           grads):
            setattr(hg, name, t.data_ptr())
        dg = torch.empty_like(g)
        scratch = _scratch(dev, lib.eagcn_head_scratch_bytes(C.byref(hp), B))
        host_seed, seed_dev = ctx.seed
        L.check(lib.eagcn_head_backward(C.byref(hp), B, ctx.training, host_seed, C.c_void_p(seed_dev or 0), _ptr(g), _ptr(saved),
                                        saved.numel(), _ptr(scratch), scratch.numel(), _ptr(dout), _ptr(dgr), C.byref(hg),
                                        _ptr(dg), _stream()), 'eagcn_head_backward')
        return (None, None, None, None, dg, *grads)


def head_forward(mods, g, training, seed, dropout):
    """out, graph_representation of the head modules ``mods`` on the fingerprints ``g``; ``seed``: an int or a 1-element int64
    device tensor (graph mode)."""
    params = [getattr(mods[mn], pn) for mn, pn in HEAD_PARAMS]
    return _HeadFn.apply(mods, training, seed, dropout, g, *params)
