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
    if r.returncode != 0 or 'DIST_WORLD1_OK' not in r.stdout:
        _dump('dist_world1_rc%d' % r.returncode, r)
    assert r.returncode == 0 and 'DIST_WORLD1_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def _run_multi(world, port):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    script = os.path.join(ROOT, 'tests', 'dist_multi_check.py')
    if world == 1:
        env.update(EAGCN_FORCE_DIST='1', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
        cmd = [sys.executable, script]
    else:
        for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
            env.pop(k, None)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
               '--master-addr', '127.0.0.1', '--master-port', str(port), script]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=840)
    if r.returncode != 0 or r.stdout.count('DIST_MULTI_OK') != world:
        _dump('dist_multi_world%d_rc%d' % (world, r.returncode), r)
    assert r.returncode == 0 and r.stdout.count('DIST_MULTI_OK') == world, (r.stdout[-3000:], r.stderr[-4000:])
    return r.stdout


@pytest.mark.timeout(900)
def test_host_issued_allreduce_fallback_averages_before_the_update():
    """EAGCN_COMM_IN_GRAPH=0: the collective is never captured; every step (a slot's first eager step included, where no .grad is
    attached yet) must average the whole flat gradient buffer exactly once before FlatAdam consumes it."""
    import torch
    n = min(torch.cuda.device_count(), 4)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29561', HSA_ENABLE_IPC_MODE_LEGACY='0', EAGCN_COMM_IN_GRAPH='0')
    script = os.path.join(ROOT, 'tests', 'dist_fallback_check.py')
    if n < 2:
        n = 1
        env.update(EAGCN_FORCE_DIST='1', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
        cmd = [sys.executable, script]
    else:
        for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
            env.pop(k, None)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
               '--master-port', '29561', script]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=840)
    if r.returncode != 0 or r.stdout.count('DIST_FALLBACK_OK') != n:
        _dump('dist_fallback_rc%d' % r.returncode, r)
    assert r.returncode == 0 and r.stdout.count('DIST_FALLBACK_OK') == n, (r.stdout[-3000:], r.stderr[-4000:])


def _dump(tag, r):
    """Keep the complete outputs of a failed rank launch (pytest elides long assertion messages)."""
    d = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, tag + '.txt'), 'w') as f:
            f.write('---- stdout ----\n' + r.stdout + '\n---- stderr ----\n' + r.stderr)
    except OSError:
        pass


@pytest.mark.timeout(900)
def test_data_parallel_step_world1_in_graph_allreduce_and_sync_bn():
    """The multi-rank check with ONE rank (EAGCN_FORCE_DIST=1): RCCL collectives captured inside the step graph, the 'dp' loss
    scale, the sync-BatchNorm hook between the reduction and finalize kernels -- all against the CPU oracle."""
    out = _run_multi(1, 29541)
    assert 'captured in the step graph' in out, out[-2000:]


@pytest.mark.timeout(900)
def test_data_parallel_step_multi_gpu_vs_oracle():
    """N >= 2 ranks over RCCL/xGMI (skipped on a single-GPU box): averaged gradients == the oracle on the sharded (local-BN)
    and on the concatenated (sync-BN) batch, identical on every rank, collective inside the captured step graph."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip('needs at least 2 GPUs (found %d)' % n)
    out = _run_multi(min(n, 4), 29551)
    print(out[-3000:])
