"""Drop-in module: ``from models import *`` of the reference's train.py:20 resolves to the HIP-backed
EAGCN (same constructor / forward signature / state_dict keys as eagcn_pytorch/models.py:14-121)."""
from eagcn_amd.models import EAGCN, Concate_GCN, Weighted_GCN, weights_init  # noqa: F401
