#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc SQ pass (csv).  usage: sq_counters.py <dir-with-*_counter_collection.csv> [filter]
Derived columns (MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles):
  cyc/wave = 4 * SQ_WAVE_CYCLES / SQ_WAVES;  wait% / iwait% / active% = share of SQ_WAVE_CYCLES parked (s_waitcnt, barrier) /
  issue-stalled / issuing;  mfma% = SQ_VALU_MFMA_BUSY_CYCLES / (duration * 2.4 GHz * 1024 SIMDs) when a kernel trace is present."""
import collections
import csv
import glob
import sys


def main(d, flt=''):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('void ', '').replace('eagcn::', '').split('(')[0]
        if flt and flt not in name:
            continue
        per[name][r['Counter_Name']].append(float(r['Counter_Value']))
    dur = collections.defaultdict(list)
    kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
    if kt:
        for r in csv.DictReader(open(kt[0])):
            name = r['Kernel_Name'].replace('void ', '').replace('eagcn::', '').split('(')[0]
            dur[name].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    print('%-70s %5s %8s %8s %9s %6s %6s %7s %6s %7s %8s' % ('kernel', 'n', 'us', 'waves', 'cyc/wave', 'wait%', 'iwait%', 'active%', 'mfma%', 'ldswt%', 'ldsconf%'))
    for name, c in sorted(per.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
        avg = {k: sum(v) / len(v) for k, v in c.items()}
        n = max(len(v) for v in c.values())
        us = sum(dur[name]) / len(dur[name]) if dur.get(name) else float('nan')
        wc = avg.get('SQ_WAVE_CYCLES', float('nan'))
        waves = avg.get('SQ_WAVES', float('nan'))
        pct = lambda k: 100.0 * avg.get(k, float('nan')) / wc
        mf = 100.0 * avg.get('SQ_VALU_MFMA_BUSY_CYCLES', float('nan')) / (us * 2400.0 * 1024.0) if us == us else float('nan')
        conf = 100.0 * avg.get('SQ_LDS_BANK_CONFLICT', float('nan')) / max(avg.get('SQ_LDS_IDX_ACTIVE', float('nan')), 1e-9)
        print('%-70s %5d %8.2f %8.0f %9.0f %6.1f %6.1f %7.1f %6.1f %7.1f %8.1f' % (name[:70], n, us, waves, 4 * wc / max(waves, 1), pct('SQ_WAIT_ANY'),
              pct('SQ_WAIT_INST_ANY'), pct('SQ_ACTIVE_INST_ANY'), mf, pct('SQ_WAIT_INST_LDS'), conf))


if __name__ == '__main__':
    main(*sys.argv[1:3])
