#!/usr/bin/env python3
"""Headline benchmark: molecules/sec, forward + backward, 2-layer 5-view Concate EAGCN on a
Tox21-shaped synthetic batch (BASELINE.json metric; SURVEY.md 8d measurement plan).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch resident in HBM: batch index build from the
reference's dense collate tensors, all graph-conv layers forward, read-out + head, the train.py loss,
full backward (gradients of every parameter), and for N > 1 the gradient all-reduce.  The optimizer
is excluded.  Weak scaling: every rank processes its own batch of `--batch` molecules.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # BASELINE.json configs[1]: Tox21 12-task, 2-layer 5-view Concate, batch 256 (fp32 here)
    'tox21_c2': dict(structure='Concate', n_layers=2, widths1=[80] * 5, widths2=[140] * 5, dens=(256, 64),
                     nclass=12, n_bfeat=28, batch=256, n_max=132, n_med=16, task='class'),
    # configs[2]: HIV, 2-layer Weighted_sum, batch 1024, N=222
    'hiv_c3': dict(structure='Weighted_sum', n_layers=2, widths1=[100] * 5, widths2=[250] * 5, dens=(512, 128),
                   nclass=1, n_bfeat=28, batch=1024, n_max=222, n_med=23, task='class'),
    # configs[3]: Lipophilicity regression, 3-layer Concate, batch 4096 over 8 GPUs (512 / GPU)
    'lipo_c4': dict(structure='Concate', n_layers=3, widths1=[60] * 5, widths2=[100] * 5, dens=(128, 64),
                    nclass=1, n_bfeat=18, batch=512, n_max=115, n_med=27, task='reg'),
    # configs[4]: synthetic roofline stress, N = 256 atoms in every molecule, K = 8 views (channels
    # [32,4,2,2,2,2,2,2], widths 64 / 128: SURVEY.md section 8 config table), batch 8192 over 8 GPUs (1024 / GPU)
    'c5_synth': dict(structure='Concate', n_layers=2, widths1=[64] * 8, widths2=[128] * 8, dens=(256, 64),
                     nclass=1, n_bfeat=32, rel_channels=[32, 4, 2, 2, 2, 2, 2, 2], batch=1024, n_max=256, n_med=None,
                     all_full=True, task='reg'),
}


def rel_channels(cfg):
    return list(cfg.get('rel_channels') or [cfg['n_bfeat'], 4, 2, 2, 2])
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 256 CU x 2.4 GHz
PEAK_HBM_GBS = 8000.0


def build_model(cfg, dropout, device, graph=False):
    from eagcn_amd import EAGCN, weights_init
    m = EAGCN(cfg['n_bfeat'], 24, n_den1=cfg['dens'][0], n_den2=cfg['dens'][1], nclass=cfg['nclass'], dropout=dropout,
              widths1=cfg['widths1'], widths2=cfg['widths2'], rel_channels=rel_channels(cfg), structure=cfg['structure'],
              n_layers=cfg['n_layers'], atom_rep='lazy', grad_mode='direct', overlap_index=True, graph=graph)
    m.apply(weights_init)
    return m.to(device)


def algorithmic_flops(cfg, sizes):
    """per molecule, true atom count n, cheaper association per layer, bwd = 2x fwd."""
    K = len(cfg['widths1'])
    w1, w2 = list(cfg['widths1']), list(cfg['widths2'])
    if cfg['structure'] == 'Weighted_sum':
        w1, w2 = [sum(w1)] * K, [sum(w2)] * K
        f1, f2 = w1[0], w2[0]
    else:
        f1, f2 = sum(w1), sum(w2)
    w3 = [2 * w for w in w2]
    plan = [(24, w1, f1), (f1, w2, f2), (f2, w3, 2 * f2), (2 * f2, w3, 2 * f2)][:cfg['n_layers']]
    total = 0.0
    for n in sizes:
        n = float(n)
        for fin, ws, _ in plan:
            for fk in ws:
                total += min(2 * n * n * fin + 2 * n * fin * fk, 2 * n * fin * fk + 2 * n * n * fk)
        f_last = plan[-1][2]
        total += 2 * (f_last * cfg['dens'][0] + cfg['dens'][0] * cfg['dens'][1] + cfg['dens'][1] * cfg['nclass'])
    return 3.0 * total


def committed_traffic():
    """HBM-side bytes per launch of the dominant kernel (gemm_f32_pair_kernel) from the committed PMC passes of this command
    (tools/pmc_traffic.py -> profiles/r01_pmc_traffic.json; rocprofv3 cannot run inside bench.py)."""
    path = os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')
    try:
        table = json.load(open(path))
    except Exception:
        return None, None
    big = [v for k, v in table.items() if 'gemm_f32_pair_kernel' in k]
    n = sum(v['launches'] for v in big)
    if not n:
        return None, None
    return sum(v['hbm_bytes_per_launch'] * v['launches'] for v in big) / n, 'profiles/r01_pmc_traffic.json'


def cpu_baseline(cfg, mb, dropout, bce_w, steps=5, warmup=2):
    """The CPU oracle (oracle/eagcn_ref.py, kind 'port') timed on this host, same workload."""
    from oracle.eagcn_ref import RefEAGCN, classification_loss, regression_loss, weights_init_
    avail = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        pass
    model = RefEAGCN(cfg['n_bfeat'], 24, cfg['widths1'], cfg['widths2'], cfg['dens'][0], cfg['dens'][1],
                     cfg['nclass'], dropout, structure=cfg['structure'], n_layers=cfg['n_layers'],
                     rel_channels=rel_channels(cfg))
    weights_init_(model)
    dense = mb.dense('cpu')
    labels = torch.from_numpy(mb.labels)

    def one_step():
        model.zero_grad(set_to_none=True)
        out, _, _ = model(*dense)
        loss = classification_loss(out, labels, bce_w) if cfg['task'] == 'class' else regression_loss(out, labels)
        loss.backward()

    # the many small ATen ops of this path do not scale to hundreds of threads: pick the fastest of
    # a few thread counts (one step each after one warm-up) and report that count as `cores`
    best = (None, 1e30)
    for nt in [c for c in (8, 16, 32, 64) if c <= avail] or [avail]:
        torch.set_num_threads(nt)
        one_step()
        t0 = time.perf_counter()
        one_step()
        dt = time.perf_counter() - t0
        if dt < best[1]:
            best = (nt, dt)
    cores = best[0]
    torch.set_num_threads(cores)
    ts = []
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        model.zero_grad(set_to_none=True)
        out, _, _ = model(*dense)
        loss = classification_loss(out, labels, bce_w) if cfg['task'] == 'class' else regression_loss(out, labels)
        loss.backward()
        if i >= warmup:
            ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    model_name = ''
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    model_name = line.split(':', 1)[1].strip()
                    break
    except Exception:
        pass
    return {'value': mb.B / med, 'unit': 'molecules/s', 'cores': cores, 'kind': 'port',
            'sample': '%d timed fwd+bwd steps (median, after %d warm-up) of the same %d-molecule batch, '
                      'oracle/eagcn_ref.py RefEAGCN on torch %s CPU, %d threads, %s'
                      % (steps, warmup, mb.B, torch.__version__, cores, model_name or 'unknown CPU')}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='tox21_c2', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=None, help='molecules per GPU (default: the config\'s)')
    ap.add_argument('--dropout', type=float, default=0.3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--eager', action='store_true', help='eager launches instead of HIP-graph replay')
    ap.add_argument('--cpu-steps', type=int, default=5)
    ap.add_argument('--eval-throughput', action='store_true',
                    help='also time the eval-mode forward (forward-only graph under no_grad); reported as an extra field')
    ap.add_argument('--input', default='dense', choices=('dense', 'compact'),
                    help="dense: the reference's collate tensors (headline); compact: bond list via forward_compact")
    args = ap.parse_args()

    from eagcn_amd import _lib
    from eagcn_amd.losses import fused_classification_loss, fused_regression_loss
    from eagcn_amd.parallel import GradientAllReducer, init_distributed
    from eagcn_amd.synthetic import bce_weights, make_batch
    lib = _lib.load()                                    # fail loudly if the HIP library is missing
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: there is no CPU path')
    rank, world, local = init_distributed()
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)' % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    cfg = dict(WORKLOADS[args.workload])
    B = args.batch or cfg['batch']
    torch.manual_seed(1234 + rank)
    mb = make_batch(B=B, n_max=cfg['n_max'], n_med=cfg['n_med'], rel_channels=rel_channels(cfg),
                    seed=1234 + rank, n_tasks=cfg['nclass'], task=cfg['task'], all_full=cfg.get('all_full', False))
    dense = mb.dense(dev) if args.input == 'dense' else None
    compact = mb.compact(dev) if args.input == 'compact' else None
    labels = torch.from_numpy(mb.labels).to(dev)
    bce_w = bce_weights(cfg['nclass'])
    bce_w_dev = torch.tensor(bce_w, dtype=torch.float32, device=dev)
    model = build_model(cfg, args.dropout, dev, graph=not args.eager)
    model.train()
    reducer = GradientAllReducer(model.parameters(), model=model)
    params = list(model.parameters())

    def step():
        for p in params:            # optimizer.zero_grad(set_to_none=True) of the reference loop (train.py:317)
            p.grad = None
        out, _, _ = model(*dense) if compact is None else model.forward_compact(*compact)
        if cfg['task'] == 'class':
            loss = fused_classification_loss(out, labels, bce_w_dev)
        else:
            loss = fused_regression_loss(out, labels)
        loss.backward()
        reducer()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    profile_in_loop = args.eager            # HIP events cannot bracket kernels inside a replayed graph
    lib.eagcn_prof_reset()
    lib.eagcn_prof_enable(1 if profile_in_loop else 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    lib.eagcn_prof_enable(0)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if not torch.isfinite(loss.detach()).item():
        raise SystemExit('non-finite loss')
    prof_steps = args.steps
    if not profile_in_loop:
        # per-kernel-class durations: the same step, same batch, eager launches with one HIP-event pair per
        # kernel class on the launch stream (the timed region above replays the identical kernels as graphs)
        model.graph = False
        prof_steps = min(args.steps, 30)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        lib.eagcn_prof_reset()
        lib.eagcn_prof_enable(1)
        for _ in range(prof_steps):
            step()
        torch.cuda.synchronize()
        lib.eagcn_prof_enable(0)
        model.graph = not args.eager

    # per-kernel-class time from HIP events recorded on the launch stream inside the timed region
    kern = {}
    for tag in range(lib.eagcn_prof_ntags()):
        ms, work, n = C.c_double(), C.c_double(), C.c_int64()
        lib.eagcn_prof_read(tag, C.byref(ms), C.byref(work), C.byref(n))
        kern[lib.eagcn_prof_tag_name(tag).decode()] = (ms.value, work.value, n.value)
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        # dominant kernel: the paired backward products of the hidden layers (dX = dP.W^T and dW = X^T.dP in one grid);
        # models without a hidden layer below the top have no such launch -> the plain layer GEMMs
        pair = kern.get('gemm_pair', (0.0, 0.0, 0))[2] > 0
        g_ms, g_work, g_n = kern['gemm_pair'] if pair else kern['gemm']
        achieved = (g_work / (g_ms * 1e-3)) / 1e12 if g_ms > 0 else 0.0
        traffic, traffic_src = committed_traffic() if args.workload == 'tox21_c2' and B == 256 else (None, None)
        out = {
            'metric': 'molecules/sec fwd+bwd, 2-layer 5-view EAGCN, Tox21 batch' if args.workload == 'tox21_c2'
                      else 'molecules/sec fwd+bwd, EAGCN %s' % args.workload,
            'value': round(value, 1), 'unit': 'molecules/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_step, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '%s: %s, %d-layer %d-view, widths %s/%s, %d tasks, batch %d per GPU, '
                                   'N_pad %d, dropout %.2f, loss %s' %
                                   (args.workload, cfg['structure'], cfg['n_layers'], len(cfg['widths1']),
                                    cfg['widths1'][0], cfg['widths2'][0], cfg['nclass'], B, mb.N, args.dropout,
                                    'weighted BCE' if cfg['task'] == 'class' else 'MSE'),
                       'input': args.input, 'global_batch': world * B, 'atoms_per_batch': int(mb.sizes.sum()),
                       'parallelism': 'dp%d' % world},
            'algorithmic_gflop_per_step': round(algorithmic_flops(cfg, mb.sizes) / 1e9, 3),
            'roofline': {'kernel': 'gemm_f32_pair_kernel<64, 64, 16, 4, false> (dX = dP.W^T and dW = X^T.dP of a layer in one grid, fp32 MFMA)'
                                   if pair else 'gemm_f32_kernel (flat X.[W_1..W_K] transform and its backward products)',
                         'bound': 'mfma', 'achieved': round(achieved, 3), 'peak': PEAK_FP32_MFMA_TFLOPS,
                         'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                         'traffic': None if traffic is None else round(traffic),
                         'traffic_note': None if traffic is None else
                         'bytes per launch of gemm_f32_pair_kernel, memory-side FETCH_SIZE x2 + WRITE_SIZE from %s; '
                         'algorithmic operand bytes are 44.6 MB (dP 13.5, X 7.7, W 1.1 read; dX 7.7 and 13 split-K '
                         'partials of dW 14.6 written)' % traffic_src,
                         'launches': int(g_n), 'avg_launch_us': round(g_ms * 1e3 / max(g_n, 1), 3),
                         'measured': 'HIP events on the launch stream, %s' % ('inside the timed region' if profile_in_loop else
                                     '%d eager steps of the same workload right after the timed graph-replay region' % prof_steps)},
            'layer_gemm_tflops_all': round(((kern['gemm'][1] + kern.get('gemm_pair', (0, 0, 0))[1]) /
                                            max((kern['gemm'][0] + kern.get('gemm_pair', (0, 0, 0))[0]) * 1e-3, 1e-12)) / 1e12, 3),
            'kernel_ms_per_step': {k: round(v[0] / prof_steps, 4) for k, v in kern.items()},
            'execution': 'eager launches' if args.eager else 'HIP graph replay (forward + backward), eager batch index',
        }
        if args.eval_throughput and not args.eager:
            model.eval()
            with torch.no_grad():
                for _ in range(10):
                    model(*dense) if compact is None else model.forward_compact(*compact)
                torch.cuda.synchronize()
                te = time.perf_counter()
                for _ in range(args.steps):
                    model(*dense) if compact is None else model.forward_compact(*compact)
                torch.cuda.synchronize()
                te = time.perf_counter() - te
            model.train()
            out['eval_forward'] = {'molecules_per_s': round(B * args.steps / te, 1), 'ms_per_batch': round(te / args.steps * 1e3, 4),
                                   'execution': 'forward-only HIP graph, eval mode, no_grad'}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(cfg, mb, args.dropout, bce_w, steps=args.cpu_steps)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
