#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; csv output, one directory each) of
`bench.py --eager` into a per-kernel HBM-traffic table.

MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide coalesced streaming read (requests tallied at 64 B instead of 128 B), so
it is doubled; WRITE_SIZE is used as reported (uncalibrated).  Infinity-Cache hits appear to be
counted, i.e. this is memory-side (fabric) traffic, an upper bound of true HBM traffic.

    python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE profiles/r01_pmc_traffic
"""
import collections
import csv
import json
import sys


def load(d, counter):
    rows = [r for r in csv.DictReader(open('%s/t_counter_collection.csv' % d)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r.get('Dispatch_Id', 0) or 0))
    per = collections.defaultdict(list)
    nth = collections.Counter()
    for r in rows:
        name = r['Kernel_Name'].replace('void ', '').replace('eagcn::', '').split('(')[0]
        grid = int(r['Grid_Size']) // max(int(r['Workgroup_Size']), 1)
        tag = 'big' if ('gemm' in name and grid >= 256) else ''
        if 'bx3_kernel' in name:
            # the plane GEMM is ONE kernel for the forward product and for the dX + dW pair: in a 2-layer model's step its launches
            # alternate (forward of layer 2, then the pair of layer 2) -- labelled by their order of dispatch
            tag = 'forward product' if nth[name] % 2 == 0 else 'dX + dW pair'
            nth[name] += 1
        per[(name, tag)].append(float(r['Counter_Value']))
    return per


def main(fdir, wdir, out):
    f, w = load(fdir, 'FETCH_SIZE'), load(wdir, 'WRITE_SIZE')
    table = {}
    for key in f:
        n = len(f[key])
        fetch = sum(f[key]) / n * 1024 * 2            # KiB -> bytes, gfx950 half-count correction
        write = sum(w.get(key, [0.0])) / max(len(w.get(key, [0.0])), 1) * 1024
        label = key[0] + ((' [>=256 workgroups]' if key[1] == 'big' else ' [%s]' % key[1]) if key[1] else '')
        table[label] = {'launches': n, 'read_bytes_per_launch': fetch, 'write_bytes_per_launch': write,
                        'hbm_bytes_per_launch': fetch + write}
    json.dump(table, open(out + '.json', 'w'), indent=1, sort_keys=True)
    with open(out + '.txt', 'w') as fh:
        fh.write('# HBM-side traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of\n')
        fh.write('#   python bench.py --steps 10 --warmup 3 --no-cpu-baseline --eager\n')
        fh.write('# FETCH_SIZE doubled (gfx950 half-count), KiB -> bytes; WRITE_SIZE as reported.\n')
        fh.write('%-62s %8s %12s %12s\n' % ('kernel', 'launches', 'read MB', 'write MB'))
        for k, v in sorted(table.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches']):
            fh.write('%-62s %8d %12.2f %12.2f\n' % (k[:62], v['launches'], v['read_bytes_per_launch'] / 1e6,
                                                     v['write_bytes_per_launch'] / 1e6))
    print(open(out + '.txt').read())


if __name__ == '__main__':
    main(*sys.argv[1:4])
