"""eagcn_amd: MI355X-native EAGCN forward/backward hot path (hand-written HIP behind a C ABI)."""
from .layers import (GAT, AFM_BatchNorm, Ave_multi_view, Dense, Diff_Pooling, GraphAttentionLayer,  # noqa: F401
                     GraphConv_base, GraphConv_block, GraphConv_Layer, Vanilla_GCN)
from .models import EAGCN, Concate_GCN, Weighted_GCN, weights_init  # noqa: F401

__all__ = ['EAGCN', 'Concate_GCN', 'Weighted_GCN', 'GraphConv_Layer', 'GraphConv_block', 'GraphConv_base',
           'AFM_BatchNorm', 'Ave_multi_view', 'Dense', 'Vanilla_GCN', 'GAT', 'GraphAttentionLayer', 'Diff_Pooling',
           'weights_init']
