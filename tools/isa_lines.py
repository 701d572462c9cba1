#!/usr/bin/env python3
"""Static instruction profile of one kernel by source line: compile a .hip with line tables (-gline-tables-only -S
--cuda-device-only) and count the ISA instructions (VALU / SALU / LDS / memory) each source line of the given file produced.
Usage: python tools/isa_lines.py <asm.s> <kernel symbol substring> <source file> [first_line last_line]"""
import collections
import re
import sys


def main():
    asm, sym, srcpath = sys.argv[1:4]
    lo, hi = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (0, 10 ** 9)
    src = open(srcpath).read().split('\n')
    lines = open(asm).read().split('\n')
    start = next(i for i, l in enumerate(lines) if sym in l and l.split(';')[0].strip().endswith(':') and not l.startswith(('\t', '.', ';')))
    end = start
    while not lines[end].strip().startswith('s_endpgm'):
        end += 1
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2))
    cur = (None, None)
    kinds = collections.defaultdict(collections.Counter)
    for l in lines[start:end]:
        m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        t = l.strip()
        if not t or t.startswith(('.', ';', '//')) or t.endswith(':'):
            continue
        op = t.split()[0]
        k = 'V' if op.startswith('v_') else 'S' if op.startswith('s_') else 'L' if op.startswith('ds_') else 'M'
        if op.startswith('v_') and '_f64' in op:
            k = 'V64'
        kinds[cur][k] += 1
    byfile = collections.Counter()
    for (f, ln), c in kinds.items():
        byfile[files.get(f, '?')] += sum(c.values())
    print('instructions by file:', byfile.most_common(6))
    base = srcpath.split('/')[-1]
    fid = [f for f, n in files.items() if n.endswith(base)]
    tot = collections.Counter()
    for (f, ln), c in sorted(kinds.items(), key=lambda x: (x[0][1] or 0)):
        if f in fid and lo <= ln <= hi:
            tot.update(c)
            print(f'{ln:5d} V{c["V"]:4d} V64{c["V64"]:4d} S{c["S"]:4d} L{c["L"]:3d} M{c["M"]:3d} | {src[ln - 1].strip()[:120]}')
    print('total', dict(tot))


if __name__ == '__main__':
    main()
