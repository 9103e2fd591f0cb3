"""The data-parallel path on real hardware with one rank: torch.distributed over RCCL (backend "nccl" on ROCm) is
initialised, the in-place AVG all-reduce runs on the flat gradient buffer of a graph-mode model and the 1-element
collective of the global BCE normalisation runs.  (Multi-GPU runs are the driver's; world_size-2 logic is covered on CPU
with gloo in test_parallel_cpu.py.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_rccl_world1_flat_buffer_allreduce_in_graph_mode():
    env = dict(os.environ, EAGCN_FORCE_DIST='1', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
               MASTER_PORT='29531', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dist_world1_check.py')], env=env, capture_output=True,
                       text=True, timeout=540)
    assert r.returncode == 0 and 'DIST_WORLD1_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
