#!/usr/bin/env python3
"""Several consecutive steady-state steps of a rocprofv3 --kernel-trace result with the QUEUE every dispatch ran on: which kernels
of the side stream (the next batch's index build) run under which kernels of the step, and where the main stream idles.

    python tools/rocprof_streams.py gpurun_out/prof/x_results.db [steps=3] > profiles/r06_streams.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    return name.replace('void ', '').replace('eagcn::', '')[:70]


def main(path, steps='3', anchor='pack_params'):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    sel = 'select name, start, end, %s from kernels' % (qcol or '0')
    ev = sorted((s, e, short(n), q) for n, s, e, q in cur.execute(sel))
    anchors = [i for i, x in enumerate(ev) if anchor in x[2]]
    a = anchors[len(anchors) // 3]
    b = anchors[len(anchors) // 3 + int(steps)]
    t0 = ev[a][0]
    queues = sorted({x[3] for x in ev[a:b]})
    print('# %s consecutive steps of %s (anchor %r); queues %s; columns: start us, duration us, queue, kernel' % (steps, path, anchor, queues))
    last_end = {}
    for s, e, n, q in ev[a:b]:
        idle = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        pad = '' if q == queues[0] else ' ' * 40
        print('%9.2f %8.2f  q%-3s %s%s%s' % ((s - t0) / 1e3, (e - s) / 1e3, q, pad, n, '   (queue idle %.1f)' % idle if idle > 4 else ''))
        last_end[q] = e


if __name__ == '__main__':
    main(*sys.argv[1:])
