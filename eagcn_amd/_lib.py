"""ctypes binding of libeagcn_hip.so (the C ABI declared in include/eagcn_hip.h).

This is the reference-side stub a maintainer would add: plain pointers and sizes,
no torch types cross the boundary.  There is no CPU fallback: if the library is missing or fails
to load, importing the ops raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libeagcn_hip.so')

MAX_VIEWS = 8
MAX_SEGS = 8
META_WORDS = 16
META_NLOG = 8
META_T, META_NMAX, META_NTILES, META_BAD_ADJ, META_BAD_REL, META_NEDGE, META_OVERFLOW, META_EDGE_OVERFLOW = range(8)
META_NBLK = 9
STRUCT_CONCATE, STRUCT_WEIGHTED = 0, 1

_fp = C.c_void_p   # device pointers travel as integers


class Batch(C.Structure):
    _fields_ = [('B', C.c_int32), ('N', C.c_int32), ('K', C.c_int32), ('ldc', C.c_int32),
                ('T', C.c_int32), ('n_max', C.c_int32), ('n_tiles', C.c_int32),
                ('channels', C.c_int32 * MAX_VIEWS),
                ('code', _fp), ('deg_bn', _fp), ('nat', _fp), ('row0', _fp), ('tile0', _fp),
                ('meta', _fp), ('row_mol', _fp), ('row_loc', _fp), ('row_m', _fp), ('row_deg', _fp),
                ('tile_mol', _fp), ('row_info', _fp), ('tile_info', _fp),
                ('rel_vec', _fp * MAX_VIEWS), ('rel_c', C.c_int32 * MAX_VIEWS),
                ('E', C.c_int32), ('n_logical', C.c_int32), ('ecnt', _fp), ('edge0', _fp), ('mol_info', _fp), ('row_ptr', _fp),
                ('col_ptr', _fp), ('nbr', _fp), ('tnbr', _fp), ('ecode', _fp), ('tcode', _fp),
                ('build_lists', C.c_int32), ('t_hint', C.c_int32), ('blk', _fp)]


def bond_ptrs_len(T, B):
    return 4 * T + 4 * B + 8 * (B + 1) + 8


def set_bond_lists(c, small_ptr, ptrs, edges, E):
    """Point the bond-list fields of a Batch at their buffers: `small_ptr` = device address of [ecnt B | edge0 B+1] int32,
    `ptrs` = int32 [bond_ptrs_len(T, B)] (row_ptr | col_ptr | mol_info | blk), `edges` = int32 [6*E + 2]
    (nbr | tnbr | ecode u64 | tcode u64)."""
    c.E = int(E)
    c.ecnt, c.edge0 = small_ptr, small_ptr + 4 * c.B
    pb, T = ptrs.data_ptr(), c.T
    c.row_ptr, c.col_ptr, c.mol_info = pb, pb + 8 * T, pb + 16 * T
    blk = pb + 16 * T + 16 * c.B
    c.blk = blk + (-blk) % 16 if ptrs.numel() >= bond_ptrs_len(T, c.B) else None      # (int4 records)
    eb = edges.data_ptr()
    c.nbr, c.tnbr = eb, eb + 4 * E
    code = eb + 8 * E
    code += (-code) % 8
    c.ecode, c.tcode = code, code + 8 * E


class GatParams(C.Structure):
    _fields_ = [('fin', C.c_int32), ('ld_in', C.c_int32), ('F', C.c_int32), ('training', C.c_int32),
                ('alpha', C.c_float), ('att_dropout', C.c_float), ('dropout', C.c_float), ('reserved_', C.c_float),
                ('seed', C.c_uint64), ('W', _fp), ('a', _fp), ('seed_dev', C.c_void_p)]


POOL_MAX = 8


class PoolAtt(C.Structure):
    _fields_ = [('K', C.c_int32), ('mode', C.c_int32), ('att_c', C.c_int32 * MAX_VIEWS), ('att_w', _fp * MAX_VIEWS),
                ('ave_a', _fp), ('self_r', _fp), ('datt_w', _fp * MAX_VIEWS), ('dave_a', _fp), ('dself_r', _fp)]


class Layout(C.Structure):
    _fields_ = [('nseg', C.c_int32), ('width', C.c_int32 * MAX_SEGS), ('pad', C.c_int32 * MAX_SEGS)]


class LayerParams(C.Structure):
    _fields_ = [('K', C.c_int32), ('structure', C.c_int32), ('training', C.c_int32),
                ('width', C.c_int32 * MAX_VIEWS), ('inp', Layout),
                ('dropout', C.c_float), ('bn_eps', C.c_float), ('bn_momentum', C.c_float),
                ('seed', C.c_uint64), ('seed_dev', _fp),
                ('att_w', _fp * MAX_VIEWS), ('self_r', _fp * MAX_VIEWS), ('W', _fp * MAX_VIEWS),
                ('bias', _fp * MAX_VIEWS), ('gamma', _fp * MAX_VIEWS), ('beta', _fp * MAX_VIEWS),
                ('run_mean', _fp * MAX_VIEWS), ('run_var', _fp * MAX_VIEWS), ('ave_w', _fp)]


class LayerBufs(C.Structure):
    _fields_ = [('x', _fp), ('P', _fp), ('Y', _fp), ('rscale', _fp), ('bn', _fp), ('xout', _fp),
                ('pad_row', _fp), ('scratch', _fp), ('scratch_bytes', C.c_size_t),
                ('aux_stream', _fp), ('packed', _fp), ('packed_bytes', C.c_size_t),
                ('stats_hook', _fp), ('stats_user', _fp), ('x_planes', _fp), ('xout_planes', _fp)]


class LayerGrads(C.Structure):
    _fields_ = [('dW', _fp * MAX_VIEWS), ('dbias', _fp * MAX_VIEWS), ('dgamma', _fp * MAX_VIEWS),
                ('dbeta', _fp * MAX_VIEWS), ('datt_w', _fp * MAX_VIEWS), ('dself_r', _fp * MAX_VIEWS),
                ('dave_w', _fp)]


class HeadParams(C.Structure):
    _fields_ = [('f_in', C.c_int32), ('n_den1', C.c_int32), ('n_den2', C.c_int32), ('nclass', C.c_int32),
                ('dropout', C.c_float), ('bn_eps', C.c_float), ('bn_momentum', C.c_float),
                ('den1_w', _fp), ('den2_w', _fp), ('den3_w', _fp),
                ('gbn_w', _fp), ('gbn_b', _fp), ('gbn_rm', _fp), ('gbn_rv', _fp),
                ('bn1_w', _fp), ('bn1_b', _fp), ('bn1_rm', _fp), ('bn1_rv', _fp),
                ('bn2_w', _fp), ('bn2_b', _fp), ('bn2_rm', _fp), ('bn2_rv', _fp)]


class HeadGrads(C.Structure):
    _fields_ = [(n, _fp) for n in ('d_den1_w', 'd_den2_w', 'd_den3_w', 'd_gbn_w', 'd_gbn_b', 'd_bn1_w', 'd_bn1_b',
                                   'd_bn2_w', 'd_bn2_b')]


class Model(C.Structure):
    _fields_ = [('n_layers', C.c_int32), ('molfp_mode', C.c_int32), ('training', C.c_int32),
                ('head_seed', C.c_uint64), ('head_seed_dev', _fp), ('input_packed', C.c_int32), ('aux_stream', _fp),
                ('layer', LayerParams * 4), ('head', HeadParams), ('stats_hook', _fp), ('stats_user', _fp),
                ('stats_world', C.c_int32), ('fuse_readout', C.c_int32), ('fwd_signal', _fp), ('wait_flag', _fp), ('start_signal', _fp)]


class StepLoss(C.Structure):
    _fields_ = [('kind', C.c_int32), ('labels', _fp), ('class_weight', _fp), ('loss', _fp), ('scale', _fp), ('dout', _fp)]


# int hook(double* buf, int n, void* stream, void* user): cross-rank sum in place (sync-BatchNorm, eagcn_hip.h)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)


# name -> (restype, argtypes); also the list the CPU test checks against include/eagcn_hip.h
SIGNATURES = {
    'eagcn_abi_version': (C.c_int, []),
    'eagcn_struct_size': (C.c_size_t, [C.c_int]),
    'eagcn_last_error': (C.c_char_p, []),
    'eagcn_pad16': (C.c_int, [C.c_int]),
    'eagcn_layer_out_ld': (C.c_int, [C.POINTER(LayerParams)]),
    'eagcn_layer_fp': (C.c_int, [C.POINTER(LayerParams)]),
    'eagcn_layer_packed_bytes': (C.c_size_t, [C.POINTER(Batch), C.POINTER(LayerParams)]),
    'eagcn_layer_fwd_scratch_bytes': (C.c_size_t, [C.POINTER(Batch), C.POINTER(LayerParams)]),
    'eagcn_layer_bwd_scratch_bytes': (C.c_size_t, [C.POINTER(Batch), C.POINTER(LayerParams)]),
    'eagcn_index_build': (C.c_int, [_fp, C.POINTER(_fp), C.POINTER(Batch), _fp, _fp]),
    'eagcn_index_from_bonds': (C.c_int, [_fp, _fp, _fp, _fp, C.c_int64, C.POINTER(Batch), _fp, _fp]),
    'eagcn_index_rows': (C.c_int, [C.POINTER(Batch), _fp]),
    'eagcn_set_gemm_mode': (C.c_int, [C.c_int]),
    'eagcn_stream_wait_counter': (C.c_int, [_fp, C.c_uint32, _fp]),
    'eagcn_stream_signal_flag': (C.c_int, [_fp, _fp]),
    'eagcn_stream_wait_flag': (C.c_int, [_fp, C.c_double, _fp]),
    'eagcn_stream_wait_timeouts': (C.c_int, []),
    'eagcn_stream_wait_reset': (None, []),
    'eagcn_adam_step': (C.c_int, [_fp, _fp, _fp, _fp, C.c_int64, _fp, _fp, _fp, _fp]),
    'eagcn_agg_wants_bond_lists': (C.c_int, [C.c_int, C.c_int]),
    'eagcn_agg_wants_bond_lists_for': (C.c_int, [C.c_int, C.c_int, C.c_int]),
    'eagcn_bx3_used_splits': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'eagcn_set_bx3_wide': (C.c_int, [C.c_int]),
    'eagcn_bx3_pair_used_splits': (C.c_int, [C.c_int] * 7),
    'eagcn_bx3_plane_elems': (C.c_size_t, [C.c_int, C.c_int]),
    'eagcn_bx3_split': (C.c_int, [_fp, C.c_int, C.c_int, _fp, C.c_size_t, C.c_int, C.c_int, _fp]),
    'eagcn_gemm_bx3': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_size_t, C.c_int, C.c_int, _fp, C.c_size_t, C.c_int, C.c_int,
                                 _fp, C.c_int, C.c_int, C.c_size_t, C.c_int, _fp]),
    'eagcn_gemm_bx3_pair': (C.c_int, [C.c_int, C.c_int, C.c_int, _fp, C.c_size_t, C.c_int, C.c_int, _fp, C.c_size_t, C.c_int, C.c_int, _fp, C.c_int,
                                      C.c_int, C.c_int, C.c_int, _fp, C.c_size_t, C.c_int, C.c_int, _fp, C.c_size_t, C.c_int, C.c_int, _fp, C.c_int,
                                      C.c_int, C.c_size_t, C.c_int, _fp]),
    'eagcn_pack_rows': (C.c_int, [C.POINTER(Batch), _fp, C.c_int, C.POINTER(Layout), _fp, _fp]),
    'eagcn_unpack_rows': (C.c_int, [C.POINTER(Batch), _fp, C.POINTER(Layout), _fp, _fp, C.c_int, _fp]),
    'eagcn_pad_rows': (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    'eagcn_gat_scratch_bytes': (C.c_size_t, [C.POINTER(Batch), C.c_int]),
    'eagcn_gat_forward': (C.c_int, [C.POINTER(Batch), C.POINTER(GatParams), _fp, _fp, _fp, _fp, _fp]),
    'eagcn_gat_backward': (C.c_int, [C.POINTER(Batch), C.POINTER(GatParams), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp,
                                     C.c_size_t, _fp]),
    'eagcn_pool_scratch_bytes': (C.c_size_t, []),
    'eagcn_pool_attention_forward': (C.c_int, [C.POINTER(Batch), C.POINTER(PoolAtt), _fp, C.c_int, _fp, _fp, _fp]),
    'eagcn_pool_attention_backward': (C.c_int, [C.POINTER(Batch), C.POINTER(PoolAtt), _fp, C.c_int, _fp, _fp, _fp, C.c_size_t,
                                                _fp]),
    'eagcn_pool_mix_forward': (C.c_int, [C.POINTER(Batch), C.POINTER(Layout), _fp, C.c_int, _fp, _fp, _fp, _fp, C.c_int, _fp]),
    'eagcn_pool_mix_backward': (C.c_int, [C.POINTER(Batch), C.POINTER(Layout), _fp, C.c_int, _fp, _fp, _fp, _fp, C.c_int, _fp,
                                          _fp, _fp, _fp]),
    'eagcn_pool_reduce_forward': (C.c_int, [C.POINTER(Batch), _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp]),
    'eagcn_pool_reduce_backward': (C.c_int, [C.POINTER(Batch), _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp]),
    'eagcn_layer_forward': (C.c_int, [C.POINTER(Batch), C.POINTER(LayerParams), C.POINTER(LayerBufs), _fp]),
    'eagcn_layer_backward': (C.c_int, [C.POINTER(Batch), C.POINTER(LayerParams), C.POINTER(LayerBufs),
                                       _fp, _fp, _fp, C.POINTER(LayerGrads), _fp]),
    'eagcn_attention_dense': (C.c_int, [C.POINTER(Batch), C.POINTER(LayerParams), _fp, _fp]),
    'eagcn_readout_forward': (C.c_int, [C.POINTER(Batch), _fp, C.POINTER(Layout), _fp, _fp, C.c_int, _fp,
                                        C.c_int, _fp]),
    'eagcn_readout_backward': (C.c_int, [C.POINTER(Batch), _fp, C.POINTER(Layout), _fp, C.c_int, C.c_int,
                                         _fp, _fp, _fp]),
    'eagcn_gemm_f32': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int,
                                 _fp, C.c_int, _fp]),
    'eagcn_gemm_f32_dev': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int,
                                     _fp, C.c_int, _fp, C.c_int, _fp]),
    'eagcn_gemm_sk_workspace_bytes': (C.c_size_t, []),
    'eagcn_gemm_f32_sk': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int,
                                    _fp, C.c_int, _fp, C.c_size_t, _fp]),
    'eagcn_gemm_pair_sk': (C.c_int, [C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int,
                                     C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int,
                                     _fp, C.c_size_t, _fp]),
    'eagcn_gemm_pair_sk_slabs': (C.c_int, [C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int,
                                           C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int,
                                           C.c_size_t, _fp, C.c_size_t, _fp]),
    'eagcn_gemm_sk_plan': (C.c_int, [C.c_int] * 7 + [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'eagcn_gemm_sk_timeouts': (C.c_int, []),
    'eagcn_gemm_sk_failed': (C.c_int, []),
    'eagcn_gemm_sk_reset_failed': (None, []),
    'eagcn_gemm_sk_inject_failure': (None, []),
    'eagcn_model_saved_bytes': (C.c_size_t, [C.POINTER(Batch), C.POINTER(Model)]),
    'eagcn_model_scratch_bytes': (C.c_size_t, [C.POINTER(Batch), C.POINTER(Model)]),
    'eagcn_model_atom_rep': (C.c_int, [C.POINTER(Batch), C.POINTER(Model), C.POINTER(C.c_size_t),
                                       C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    'eagcn_model_atom_rep_materialize': (C.c_int, [C.POINTER(Batch), C.POINTER(Model), _fp, C.c_size_t, _fp]),
    'eagcn_model_pack_input': (C.c_int, [C.POINTER(Batch), C.POINTER(Model), _fp, _fp, C.c_size_t, _fp]),
    'eagcn_model_forward': (C.c_int, [C.POINTER(Batch), C.POINTER(Model), _fp, _fp, _fp, C.c_size_t, _fp,
                                      C.c_size_t, _fp, _fp, _fp]),
    'eagcn_model_backward': (C.c_int, [C.POINTER(Batch), C.POINTER(Model), _fp, _fp, C.c_size_t, _fp, C.c_size_t,
                                       _fp, _fp, C.POINTER(LayerGrads), C.POINTER(HeadGrads), _fp]),
    'eagcn_model_backward_range': (C.c_int, [C.POINTER(Batch), C.POINTER(Model), _fp, _fp, C.c_size_t, _fp, C.c_size_t,
                                             _fp, _fp, C.POINTER(LayerGrads), C.POINTER(HeadGrads), C.c_int, C.c_int, C.c_int, _fp]),
    'eagcn_model_forward_step': (C.c_int, [C.POINTER(Batch), C.POINTER(Model), _fp, _fp, _fp, C.c_size_t, _fp, C.c_size_t, _fp, _fp,
                                           C.POINTER(StepLoss), _fp, C.POINTER(HeadGrads), _fp]),
    'eagcn_head_saved_bytes': (C.c_size_t, [C.POINTER(HeadParams), C.c_int]),
    'eagcn_head_scratch_bytes': (C.c_size_t, [C.POINTER(HeadParams), C.c_int]),
    'eagcn_head_forward': (C.c_int, [C.POINTER(HeadParams), C.c_int, C.c_int, C.c_uint64, _fp, _fp, _fp, C.c_size_t, _fp, C.c_size_t,
                                     _fp, _fp, _fp]),
    'eagcn_head_backward': (C.c_int, [C.POINTER(HeadParams), C.c_int, C.c_int, C.c_uint64, _fp, _fp, _fp, C.c_size_t, _fp, C.c_size_t,
                                      _fp, _fp, C.POINTER(HeadGrads), _fp, _fp]),
    'eagcn_bce_loss': (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp]),
    'eagcn_mse_loss': (C.c_int, [_fp, _fp, C.c_int, _fp, _fp, _fp]),
    'eagcn_eval_append': (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, C.c_int64, _fp]),
    'eagcn_prof_enable': (None, [C.c_int]),
    'eagcn_prof_reset': (None, []),
    'eagcn_prof_ntags': (C.c_int, []),
    'eagcn_prof_tag_name': (C.c_char_p, [C.c_int]),
    'eagcn_prof_read': (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

_lib = None


class EagcnHipError(RuntimeError):
    pass


ABI_VERSION = 7    # include/eagcn_hip.h eagcn_abi_version(): struct layouts + signatures this binding was written against


def load():
    """dlopen the HIP library (once).  Raises if it is not built -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EagcnHipError(
            'libeagcn_hip.so is not built (%s). Run `python -c "import __graft_entry__ as g; g.build()"` '
            'or `python -m eagcn_amd.build`; eagcn_amd has no CPU fallback.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.eagcn_abi_version.restype = C.c_int
    if lib.eagcn_abi_version() != ABI_VERSION:
        raise EagcnHipError('libeagcn_hip.so speaks ABI version %d, this binding version %d: rebuild it '
                            '(`python -c "import __graft_entry__ as g; g.build()"`)' % (lib.eagcn_abi_version(), ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    for which, cls in ((0, Batch), (1, Layout), (2, LayerParams), (3, LayerBufs), (4, LayerGrads), (5, HeadParams), (6, HeadGrads),
                       (7, Model), (10, StepLoss)):
        if lib.eagcn_struct_size(which) != C.sizeof(cls):
            raise EagcnHipError('ABI mismatch: %s is %d bytes here, %d in libeagcn_hip.so'
                                % (cls.__name__, C.sizeof(cls), lib.eagcn_struct_size(which)))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().eagcn_last_error()
        raise EagcnHipError('%s failed (%d): %s' % (what, rc, msg.decode() if msg else '?'))


def pad16(w):
    return (int(w) + 15) & ~15


def pad4(w):
    return (int(w) + 3) & ~3


def make_layout(widths, pads):
    lay = Layout()
    lay.nseg = len(widths)
    for i, (w, p) in enumerate(zip(widths, pads)):
        lay.width[i] = int(w)
        lay.pad[i] = int(p)
    return lay
