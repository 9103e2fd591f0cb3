#!/usr/bin/env python3
"""How much of a fused step's period is NOT its kernels?  (a) the normal loop (index build of the next batch on the side stream,
cross-stream wait, graph replay); (b) the two captured step graphs replayed back to back with nothing in between (same resident
batch: kernel chain + bare graph-to-graph gap); (c) the same with an event record + wait on an idle stream's event between replays."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from eagcn_amd.synthetic import bce_weights, make_batch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = dict(bench.WORKLOADS['tox21_c2'])
dev = torch.device('cuda', 0)
mb = make_batch(B=B, n_max=cfg['n_max'], n_med=cfg['n_med'], rel_channels=(28, 4, 2, 2, 2), seed=1234, n_tasks=12)
dense = mb.dense(dev)
labels = torch.from_numpy(mb.labels).to(dev)
bw = torch.tensor(bce_weights(12), dtype=torch.float32, device=dev)
model = bench.build_model(cfg, 0.3, dev, graph=True).train()
params = list(model.parameters())


def step():
    for p in params:
        p.grad = None
    return model.fused_step(dense, labels, 'class', bw, None)[0]


for _ in range(20):
    step()
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print('(a) normal loop            %.1f us/step' % ((time.perf_counter() - t0) / n * 1e6))
import ctypes as C  # noqa: E402
from eagcn_amd import _lib as L  # noqa: E402
from eagcn_amd.ops import _index_stream  # noqa: E402
lib = L.load()
runner = next(iter(model._runners.values()))
g = [runner.graphs[0][2], runner.graphs[1][2]]
side = _index_stream(dev)
main = torch.cuda.current_stream(dev)


def ready(slot):          # what _prepare ends with: the step graph's first launch polls this flag
    if runner.use_flag:
        lib.eagcn_stream_signal_flag(C.c_void_p(runner.ready.data_ptr() + 4 * slot), C.c_void_p(side.cuda_stream))


def run(tag, between):
    for i in range(10):
        ready(i & 1); between(i); g[i & 1].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        ready(i & 1)
        between(i)
        g[i & 1].replay()
    torch.cuda.synchronize()
    print('%-44s %.1f us/step' % (tag, (time.perf_counter() - t0) / n * 1e6))


def ev_wait(i):
    ev = torch.cuda.Event()
    ev.record(side)
    main.wait_event(ev)


def ev_record_main(i):
    torch.cuda.Event().record(main)


keep = []


def ev_record_main_side_wait(i):
    ev = torch.cuda.Event()
    ev.record(main)
    side.wait_event(ev)


scalar = torch.zeros((), device=dev)
run('(b) bare replays', lambda i: None)
run('(c) + record(side), main.wait_event', ev_wait)
run('(e) + record(main)', ev_record_main)
run('(f) + record(main), side.wait_event', ev_record_main_side_wait)
run('(g) + clone of a scalar on main', lambda i: scalar.clone())
run('(b) bare replays again', lambda i: None)
