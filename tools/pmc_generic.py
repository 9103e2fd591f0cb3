#!/usr/bin/env python3
"""Per-kernel averages of every counter of one rocprofv3 --pmc pass (csv): pmc_generic.py <dir> [filter]"""
import collections
import csv
import glob
import sys


def main(d, flt=''):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    names = []
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('void ', '').replace('eagcn::', '').split('(')[0]
        if flt and flt not in name:
            continue
        per[name][r['Counter_Name']].append(float(r['Counter_Value']))
        if r['Counter_Name'] not in names:
            names.append(r['Counter_Name'])
    dur = collections.defaultdict(list)
    kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
    if kt:
        for r in csv.DictReader(open(kt[0])):
            name = r['Kernel_Name'].replace('void ', '').replace('eagcn::', '').split('(')[0]
            dur[name].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print('%-64s %5s %9s ' % ('kernel', 'n', 'us') + ' '.join('%16s' % n[:16] for n in names))
    for name, c in sorted(per.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
        n = max(len(v) for v in c.values())
        us = sum(dur[name]) / len(dur[name]) if dur.get(name) else float('nan')
        print('%-64s %5d %9.2f ' % (name[:64], n, us) + ' '.join('%16.4g' % (sum(c[k]) / max(len(c[k]), 1)) if k in c else '%16s' % '-' for k in names))


if __name__ == '__main__':
    main(*sys.argv[1:3])
