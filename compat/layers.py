"""Drop-in module: put this directory first on sys.path and the reference's ``from layers import *``
(eagcn_pytorch/models.py:3, train.py) resolves to the HIP-backed classes."""
from eagcn_amd.layers import *  # noqa: F401,F403
from eagcn_amd.layers import (AFM_BatchNorm, Ave_multi_view, Dense, GraphConv_base,  # noqa: F401
                              GraphConv_block, GraphConv_Layer)
