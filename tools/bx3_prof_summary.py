#!/usr/bin/env python3
"""Per-kernel averages of every counter the passes of tools/bx3_prof.sh collected (+ durations from the kernel traces)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)


def short(n):
    return n.replace('void ', '').replace('eagcn::', '').split('(')[0][:60]


for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        per[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob(d + '/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for name in sorted(per, key=lambda k: -sum(dur.get(k, [0]))):
    us = sorted(dur.get(name, [0.0]))
    print('%s   launches %d   median %.1f us  min %.1f us' % (name, len(us), us[len(us) // 2], us[0]))
    c = per[name]
    for k in sorted(c):
        v = c[k]
        print('    %-34s avg %16.1f   (n=%d)' % (k, sum(v) / len(v), len(v)))
