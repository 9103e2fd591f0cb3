#!/usr/bin/env python3
"""Summarise the basic blocks of one kernel in a hipcc -save-temps .s file: for every block with matrix instructions, the ordered
stream of MFMA / LDS read / LDS-DMA / wait / barrier / scratch events, run-length compressed.
    python tools/isa_blocks.py file.s kernel_name_substring [min_mfma]"""
import re
import sys


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        if re.match(r'^[A-Za-z_][\w$.]*:', l) and name in l:
            start = i
            break
    if start is None:
        raise SystemExit('kernel not found')
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    blocks, cur, label = [], [], 'entry'
    for l in lines[start + 1:end + 1]:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            blocks.append((label, cur))
            label, cur = m.group(1), []
        else:
            cur.append(l.strip())
    blocks.append((label, cur))
    for label, body in blocks:
        n_mfma = sum('v_mfma' in x for x in body)
        if n_mfma < min_mfma:
            continue
        ev = []
        for x in body:
            op = x.split()[0] if x.split() else ''
            if 'v_mfma' in op: e = 'M'
            elif op.startswith('ds_read'): e = 'R'
            elif op.startswith('ds_write'): e = 'W'
            elif op.startswith('buffer_load') and 'lds' in x: e = 'D'
            elif op.startswith('scratch_') or (op.startswith('buffer_') and 'offen' in x and 'lds' not in x and 's[0:3]' in x): e = 'S!'
            elif op.startswith('global_store') or op.startswith('buffer_store'): e = 'st'
            elif op.startswith('global_load') or op.startswith('buffer_load'): e = 'ld'
            elif op == 's_waitcnt': e = '[' + ' '.join(x.split()[1:]) + ']'
            elif op == 's_barrier': e = '|B|'
            elif op == 's_setprio': e = 'p' + x.split()[1]
            elif op.startswith('s_cbranch') or op == 's_branch': e = '->' + x.split()[-1]
            elif op == 's_nop': e = 'n'
            elif op.startswith('v_'): e = 'v'
            else: continue
            ev.append(e)
        out, i = [], 0
        while i < len(ev):
            j = i
            while j < len(ev) and ev[j] == ev[i]:
                j += 1
            out.append(ev[i] + (str(j - i) if j - i > 1 else ''))
            i = j
        print('%s  (%d instr, %d mfma)\n   %s\n' % (label, len(body), n_mfma, ' '.join(out)))


if __name__ == '__main__':
    main()
