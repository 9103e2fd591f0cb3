"""nn.Module surface of the multi-view edge-attention graph convolution, backed by the HIP kernels.

Mirrors the reference's layer classes for the hot path (eagcn_pytorch/layers.py): same class names,
constructor arguments, forward signatures and parameter tree (so a reference ``state_dict`` loads
with ``strict=True`` and ``check_model.py:48-58`` style attribute walks keep working):

    GraphConv_Layer.blockK.{self_r, att.weight[1,C,1,1], graph_conv.{weight[fin,F],bias[F]},
                            batch_norm.{weight[1,1], bias[1], bn.{weight,bias,running_*}}}
    GraphConv_Layer.{self_r, ave_A.weight[K], ave.weight[K] (Weighted_sum)}

The sub-modules are parameter containers: the computation of all K blocks of a layer is ONE call
into libeagcn_hip.so (eagcn_layer_forward / eagcn_layer_backward), not a chain of ATen ops, and
there is no CPU fallback -- calling a layer with CPU tensors raises.
"""
import math

import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from . import ops
from ._lib import EagcnHipError


class GraphConv_base(nn.Module):
    """Parameter container for W [fin,fout] and bias [fout] (reference layers.py:16-50).
    Init as the reference: U(-1/sqrt(fout), 1/sqrt(fout)) (layers.py:32-36)."""

    def __init__(self, in_features, out_features, bias=False):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = Parameter(torch.empty(in_features, out_features))
        if bias:
            self.bias = Parameter(torch.empty(out_features))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        s = 1.0 / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-s, s)
        if self.bias is not None:
            self.bias.data.uniform_(-s, s)

    def extra_repr(self):
        return '%d -> %d' % (self.in_features, self.out_features)


class AFM_BatchNorm(nn.Module):
    """Container for the per-view BatchNorm1d over the feature axis of [B,N,F] (layers.py:394-412).
    ``weight``/``bias`` are the reference's extra, never-used parameters (layers.py:402-404); they
    are kept (zero-initialised instead of uninitialised) so the state_dict keys match."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, bias=True):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps, momentum, affine)
        self.weight = Parameter(torch.zeros(1, 1))
        if bias:
            self.bias = Parameter(torch.zeros(1))
        else:
            self.register_parameter('bias', None)


class Ave_multi_view(nn.Module):
    """Container for the view-mixing weights (layers.py:414-437), U(-1/sqrt(K), 1/sqrt(K))."""

    def __init__(self, ave_source_num, feature_size=0, bias=False):
        super().__init__()
        self.ave_source_num = ave_source_num
        self.weight = Parameter(torch.empty(ave_source_num))
        s = 1.0 / math.sqrt(ave_source_num)
        self.weight.data.uniform_(-s, s)


class GraphConv_block(nn.Module):
    """Parameters of one view (bond relation) of a layer (reference layers.py:52-79)."""

    def __init__(self, node_feature_in, bond_feature_num, node_feature_out, dropout):
        super().__init__()
        self.node_feature_in = node_feature_in
        self.bond_feature_num = bond_feature_num
        self.node_feature_out = node_feature_out
        self.att = nn.Conv2d(bond_feature_num, 1, kernel_size=1, stride=1, padding=0, bias=False)
        self.graph_conv = GraphConv_base(node_feature_in, node_feature_out, bias=True)
        self.batch_norm = AFM_BatchNorm(node_feature_out)
        self.dropout = dropout
        self.self_r = Parameter(torch.empty(1).uniform_(-0.01, 0.01))

    def hot_params(self):
        return (self.att.weight, self.self_r, self.graph_conv.weight, self.graph_conv.bias,
                self.batch_norm.bn.weight, self.batch_norm.bn.bias)


class GraphConv_Layer(nn.Module):
    """All views of one layer + the view merge (reference layers.py:262-325).

    forward(adjs, afms, TypeAtt, OrderAtt, AromAtt, ConjAtt, RingAtt) -> (x, A_weight), as the
    reference.  Extension: ``rel_channels`` (keyword) gives the channel count of every view and so
    allows K != 5 views; default [bond_feature_num, 4, 2, 2, 2] as hard-coded in layers.py:269-273.
    """

    def __init__(self, node_feature_in, bond_feature_num, node_out_1, node_out_2=None, node_out_3=None,
                 node_out_4=None, node_out_5=None, dropout=0.0, structure='Concate', last=False, adj_size=0,
                 *, widths=None, rel_channels=None):
        super().__init__()
        if widths is None:
            widths = [node_out_1, node_out_2, node_out_3, node_out_4, node_out_5]
        widths = [int(w) for w in widths]
        if rel_channels is None:
            rel_channels = [bond_feature_num, 4, 2, 2, 2][:len(widths)]
        if len(rel_channels) != len(widths):
            raise ValueError('rel_channels and widths must have one entry per view')
        self.K = len(widths)
        for k, (c, w) in enumerate(zip(rel_channels, widths)):
            setattr(self, 'block%d' % (k + 1), GraphConv_block(node_feature_in, c, w, dropout))
        self.node_feature_in = node_feature_in
        self.widths, self.rel_channels = widths, list(rel_channels)
        self.structure, self.last, self.dropout = structure, last, dropout
        if structure == 'Concate':
            self.total_output = sum(widths)
        elif structure == 'Weighted_sum':
            if len(set(widths)) != 1:
                raise ValueError('Weighted_sum needs equal view widths (models.py:33-47)')
            self.total_output = widths[0]
            self.ave = Ave_multi_view(self.K)
        else:
            raise ValueError("structure must be 'Concate' or 'Weighted_sum', got %r" % (structure,))
        self.ave_A = Ave_multi_view(self.K)
        self.self_r = Parameter(torch.empty(1).uniform_(-0.01, 0.01))
        self._specs = {}

    def __getstate__(self):                 # cached LayerSpec objects hold ctypes structs: rebuilt on demand
        state = super().__getstate__() if hasattr(nn.Module, '__getstate__') else self.__dict__.copy()
        state = dict(state)
        state['_specs'] = {}
        return state

    # -- packed (internal) path ------------------------------------------------------------------
    def blocks(self):
        return [getattr(self, 'block%d' % (k + 1)) for k in range(self.K)]

    def spec_for(self, in_layout):
        key = (tuple(in_layout.widths), tuple(in_layout.pads))
        sp = self._specs.get(key)
        if sp is None:
            bn = self.block1.batch_norm.bn
            sp = ops.LayerSpec(self.structure, self.widths, in_layout, self.dropout, bn.eps, bn.momentum)
            self._specs[key] = sp
        sp.dropout = float(self.dropout)
        return sp

    def forward_packed(self, index, x, in_layout, seed=None):
        """x: packed [T, in_layout.ld] -> (xout packed, pad_row, out_layout)."""
        if index.K != self.K:
            raise EagcnHipError('layer has %d views, batch has %d relation tensors' % (self.K, index.K))
        spec = self.spec_for(in_layout)
        blocks = self.blocks()
        flat = []
        for b in blocks:
            flat.extend(b.hot_params())
        buffers = [(b.batch_norm.bn.running_mean, b.batch_norm.bn.running_var) for b in blocks]
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (self.training and self.dropout > 0) else 0
        ave_w = self.ave.weight if self.structure == 'Weighted_sum' else None
        xout, pad_row = ops.layer_forward(index, spec, self.training, seed, buffers, x, ave_w, flat)
        if self.training:
            torch._foreach_add_([b.batch_norm.bn.num_batches_tracked for b in blocks], 1)
        return xout, pad_row, spec.out_layout

    # -- reference signature ---------------------------------------------------------------------
    def forward(self, adjs, afms, *rels, index=None):
        if len(rels) != self.K:
            raise EagcnHipError('expected %d relation tensors, got %d' % (self.K, len(rels)))
        if index is None:
            index = ops.BatchIndex(adjs, rels, structure={'Concate': 0, 'Weighted_sum': 1}.get(self.structure, -1))
        in_layout = ops.ColLayout.single(self.node_feature_in)
        x = ops.pack_rows(index, in_layout, afms)
        xout, pad_row, out_layout = self.forward_packed(index, x, in_layout)
        dense = ops.unpack_rows(index, out_layout, xout, pad_row if self.structure == 'Weighted_sum' else None)
        with torch.no_grad():
            a_w = ops.attention_dense(index, [b.att.weight for b in self.blocks()])
            if self.last:
                a_w = self._merge_attention(a_w, adjs)
        return dense, a_w

    def _merge_attention(self, a_w, adjs):
        """layers.py:319-324 -- only consumed by molfp_mode='pool' (outside the hot path);
        plain tensor algebra on the [K,B,N,N] stack."""
        B, N, _ = adjs.shape
        m = adjs.max(dim=2, keepdim=True)[0]
        a = (a_w * self.ave_A.weight.view(-1, 1, 1, 1)).sum(0)
        eye = torch.eye(N, device=adjs.device, dtype=adjs.dtype)
        u = torch.sigmoid(a) * adjs + torch.sigmoid(self.self_r) * (m * eye) + (1.0 - adjs) * 1e-9
        return (u / u.sum(dim=2, keepdim=True)) * m


class _Frozen:
    """Stand-in for a parameter container whose `weight` is a constant buffer (Vanilla_GCN)."""

    def __init__(self, weight):
        self.weight = weight


class Vanilla_GCN(nn.Module):
    """The Kipf-GCN baseline layer of the reference (layers.py:205-258, used by models.py:63-67 for structure='GCN'):
    A = adj + mask*I + 1e-9*(1-adj), row-normalised and row-masked, then (A.X).W + b -> BatchNorm -> relu -> dropout.
    That is one view of the edge-attention layer with every bond weight and the self weight equal to ONE and no row
    mask on the output, so it runs on the same kernels: a single-view 'Weighted_sum' layer whose attention logits are
    frozen at +40 (sigmoid(40) == 1.0f exactly) and whose mixing weight is frozen at 1.  Parameter tree as the reference:
    graph_conv.{weight,bias}, batch_norm.{weight,bias,bn.*}; the frozen constants are non-persistent buffers.
    forward(adjs, afms, TypeAtt, ...) -> (x, A) as the reference; only the first relation tensor's bond POSITIONS are
    used (= adj)."""

    structure = 'Weighted_sum'
    K = 1
    last = False

    def __init__(self, node_feature_in, node_feature_out, dropout, *, bond_channels=1):
        super().__init__()
        self.node_feature_in, self.node_feature_out = node_feature_in, node_feature_out
        self.graph_conv = GraphConv_base(node_feature_in, node_feature_out, bias=True)
        self.batch_norm = AFM_BatchNorm(node_feature_out)
        self.dropout = dropout
        self.widths, self.rel_channels = [int(node_feature_out)], [int(bond_channels)]
        self.total_output = int(node_feature_out)
        self.register_buffer('_att_w', torch.full((1, int(bond_channels), 1, 1), 40.0), persistent=False)
        self.register_buffer('_self_r', torch.full((1,), 40.0), persistent=False)
        self.register_buffer('_ave_w', torch.ones(1), persistent=False)
        self._specs = {}

    def __getstate__(self):
        state = super().__getstate__() if hasattr(nn.Module, '__getstate__') else self.__dict__.copy()
        state = dict(state)
        state['_specs'] = {}
        return state

    # the interface ModelPlan / forward_packed expect from a layer and from a view
    @property
    def att(self):
        return _Frozen(self._att_w)

    @property
    def self_r(self):
        return self._self_r

    @property
    def ave(self):
        return _Frozen(self._ave_w)

    def blocks(self):
        return [self]

    def hot_params(self):
        return (self._att_w, self._self_r, self.graph_conv.weight, self.graph_conv.bias,
                self.batch_norm.bn.weight, self.batch_norm.bn.bias)

    spec_for = GraphConv_Layer.spec_for
    forward_packed = GraphConv_Layer.forward_packed

    @property
    def block1(self):
        return self

    def forward(self, adjs, afms, *rels, index=None):
        if index is None:
            if not rels:
                raise EagcnHipError('Vanilla_GCN needs the first relation tensor (bond positions)')
            index = ops.BatchIndex(adjs, rels[:1])
        in_layout = ops.ColLayout.single(self.node_feature_in)
        x = ops.pack_rows(index, in_layout, afms)
        xout, pad_row, out_layout = self.forward_packed(index, x, in_layout)
        dense = ops.unpack_rows(index, out_layout, xout, pad_row)
        with torch.no_grad():                       # A of layers.py:250-253 (dense tensor algebra; not on the hot path)
            B, N, _ = adjs.shape
            m = adjs.max(dim=2, keepdim=True)[0]
            a = adjs + m * torch.eye(N, device=adjs.device, dtype=adjs.dtype) + (1.0 - adjs) * 1e-9
            a = (a / a.sum(dim=2, keepdim=True)) * m
        return dense, a


class GraphAttentionLayer(nn.Module):
    """Parameter container of the reference's GAT attention (layers.py:99-143): W [fin,fout], a [2*fout,1], both
    xavier_uniform(gain=1.414); attention dropout 0.5 and leaky-relu slope 0.2 are the reference's defaults (the GAT module
    of layers.py:160 does not override them)."""

    def __init__(self, in_features, out_features, dropout=0.5, alpha=0.2, concat=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.dropout, self.alpha, self.concat = dropout, alpha, concat
        self.W = Parameter(torch.zeros(in_features, out_features))
        nn.init.xavier_uniform_(self.W.data, gain=1.414)
        self.a = Parameter(torch.zeros(2 * out_features, 1))
        nn.init.xavier_uniform_(self.a.data, gain=1.414)


class GAT(nn.Module):
    """The GAT baseline layer (reference layers.py:145-203, models.py:69-73): per molecule h = X.W, attention
    softmax_j leakyrelu(a1.h_i + a2.h_j) over the bonds of atom i and i itself, attention dropout, h' = att.h, then
    dropout and relu.  The reference materialises an N x N x 2F pair tensor per molecule; csrc/gat.hip walks the bond lists
    of the batch index.  forward(adjs, afms, TypeAtt, ...) -> (x, A) as the reference (A = adjs + mask*I); the BatchNorm
    container exists in the reference's module (and state_dict) but its forward never calls it."""

    structure = 'GAT'
    K = 1
    last = False

    def __init__(self, node_feature_in, node_feature_out, dropout):
        super().__init__()
        self.node_feature_in, self.node_feature_out = node_feature_in, node_feature_out
        self.graph_conv = GraphAttentionLayer(node_feature_in, node_feature_out)
        self.batch_norm = AFM_BatchNorm(node_feature_out)
        self.dropout = dropout
        self.total_output = int(node_feature_out)

    def forward_packed(self, index, x, in_layout, seed=None):
        if len(in_layout.widths) != 1:
            raise EagcnHipError('the GAT layer takes a single-segment input layout')
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if self.training else 0
        gc = self.graph_conv
        xout = ops.gat_layer(index, self.node_feature_in, in_layout.ld, self.node_feature_out, self.training, seed,
                             self.dropout, gc.dropout, gc.alpha, x, gc.W, gc.a.view(-1))
        return xout, None, ops.ColLayout.single(self.node_feature_out, 16)

    def forward(self, adjs, afms, *rels, index=None):
        if index is None:
            if not rels:
                raise EagcnHipError('GAT needs the first relation tensor (bond positions)')
            index = ops.BatchIndex(adjs, rels[:1], bond_lists=True)
        in_layout = ops.ColLayout.single(self.node_feature_in)
        x = ops.pack_rows(index, in_layout, afms)
        xout, _, out_layout = self.forward_packed(index, x, in_layout)
        dense = ops.unpack_rows(index, out_layout, xout, None)
        with torch.no_grad():
            N = adjs.shape[1]
            m = adjs.max(dim=2, keepdim=True)[0]
            a = adjs + m * torch.eye(N, device=adjs.device, dtype=adjs.dtype)       # layers.py:189
        return dense, a


class Diff_Pooling(nn.Module):
    """Soft cluster assignment read-out (reference layers.py:492-506; models.py:90-92 builds ``pool1`` with pool_num clusters
    and an unused ``pool3``).  Parameter tree as the reference: feature_layer.weight [fin,fout], adjacent_layer.weight
    [fin,out_size], both bias-free GraphConv_base.  The computation runs on csrc/pool.hip + the flat fp32 GEMM through
    ``ops.pool_readout``; ``pooled_sum`` returns sum over clusters of relu(S^T . relu((A.X).Wf)) -- the only thing
    models.py:104-106 consumes (the pooled adjacency of layers.py:504-505 is never read and not computed)."""

    def __init__(self, in_feature, out_feature, out_size):
        super().__init__()
        if not 1 <= int(out_size) <= ops.L.POOL_MAX:
            raise ValueError('Diff_Pooling: out_size must be 1..%d' % ops.L.POOL_MAX)
        self.feature_layer = GraphConv_base(in_feature, out_feature)
        self.adjacent_layer = GraphConv_base(in_feature, out_size)

    def pooled_sum(self, index, layout, x, pad_row, last_layer):
        """x: packed output [T, ld] of ``last_layer`` (whose attention matrix A is rebuilt on the device) -> [B, fout]."""
        if isinstance(last_layer, GAT):
            kw = dict(mode='gat')
        elif isinstance(last_layer, Vanilla_GCN):
            kw = dict(mode='gcn')
        else:
            if not last_layer.last:
                raise EagcnHipError("molfp_mode='pool' needs the attention matrix of a layer built with last=True "
                                    "(layers.py:319-324)")
            kw = dict(mode='attention', att_w=[b.att.weight for b in last_layer.blocks()],
                      ave_a=last_layer.ave_A.weight, self_r=last_layer.self_r)
        return ops.pool_readout(index, layout, x, pad_row, self.feature_layer.weight, self.adjacent_layer.weight, **kw)


class Dense(nn.Module):
    """x @ W without bias by default (reference layers.py:360-392); W is U(-1/sqrt(fout), ..)."""

    def __init__(self, in_features, out_features, bias=False):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = Parameter(torch.empty(in_features, out_features))
        if bias:
            self.bias = Parameter(torch.empty(out_features))
        else:
            self.register_parameter('bias', None)
        s = 1.0 / math.sqrt(out_features)
        self.weight.data.uniform_(-s, s)
        if self.bias is not None:
            self.bias.data.uniform_(-s, s)

    def forward(self, x):
        out = ops.dense_mm(x, self.weight)           # fp32 MFMA GEMM through the C ABI, no rocBLAS
        return out + self.bias if self.bias is not None else out
