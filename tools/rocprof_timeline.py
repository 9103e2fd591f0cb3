#!/usr/bin/env python3
"""Ordered timeline of ONE steady-state step out of a rocprofv3 --kernel-trace (+ --memory-copy-trace)
result: every kernel dispatch / memory copy between two consecutive index_scan launches in the middle
of the run, with its duration and the idle gap before it.

    python tools/rocprof_timeline.py gpurun_out/prof/x_results.db [anchor-substring] > profiles/r01_x_timeline.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    return name.replace('void ', '').replace('eagcn::', '')[:90]


def main(path, anchor='index_scan'):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    ev = []
    if 'kernels' in tables:
        cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
        name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
        ev += [(s, e, short(n)) for n, s, e in cur.execute('select %s, start, end from kernels' % name_col)]
    for t in ('memory_copies', 'memory_copy'):
        if t in tables:
            cols = [r[1] for r in cur.execute('pragma table_info(%s)' % t)]
            nm = 'name' if 'name' in cols else None
            q = 'select %s, start, end from %s' % (nm or "'copy'", t)
            ev += [(s, e, '[memcpy] ' + str(n)) for n, s, e in cur.execute(q)]
            break
    ev.sort()
    anchors = [i for i, x in enumerate(ev) if anchor in x[2]]
    if len(anchors) < 4:
        raise SystemExit('anchor %r seen %d times' % (anchor, len(anchors)))
    a = anchors[len(anchors) // 3]
    b = anchors[len(anchors) // 3 + 1]
    t0 = ev[a][0]
    print('# one step of %s: events between two consecutive %r launches' % (path, anchor))
    print('%9s %8s %8s  %s' % ('t_us', 'dur_us', 'gap_us', 'event'))
    prev_end = None
    busy = 0.0
    for s, e, n in ev[a:b]:
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        print('%9.2f %8.2f %8.2f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, n))
        prev_end = max(prev_end or e, e)
        busy += (e - s) / 1e3
    print('# step span %.2f us, sum of durations %.2f us, %d events' % ((ev[b][0] - t0) / 1e3, busy, b - a))


if __name__ == '__main__':
    main(*sys.argv[1:])
