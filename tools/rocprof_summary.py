#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace result (rocpd sqlite .db) as a per-kernel table:
calls, total, average, min, max duration and share of GPU kernel time.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/r01_x.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(.*$', '', name)
    name = name.replace('void ', '').replace('eagcn::', '')
    return name[:110]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = cur.execute('select %s, start, end from kernels' % name_col).fetchall()
    agg = {}
    for n, s, e in rows:
        d = (e - s) / 1e3
        a = agg.setdefault(short(n), [0, 0.0, 1e30, 0.0, []])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
        a[4].append(d)
    total = sum(a[1] for a in agg.values())
    print('# rocprofv3 --kernel-trace summary of %s' % path)
    print('# %d dispatches, %.3f ms total GPU kernel time' % (len(rows), total / 1e3))
    print('%-112s %7s %12s %10s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'median_us', 'min_us', 'max_us', '%'))
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        med = sorted(a[4])[len(a[4]) // 2]
        print('%-112s %7d %12.1f %10.2f %10.2f %10.2f %10.2f %6.2f' % (n, a[0], a[1], a[1] / a[0], med, a[2], a[3], 100 * a[1] / total))


if __name__ == '__main__':
    main(sys.argv[1])
