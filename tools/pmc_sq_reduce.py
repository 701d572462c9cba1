#!/usr/bin/env python3
"""Per-kernel averages of the raw SQ counters collected by tools/pmc_sq.sh (second half of the launches of each kernel)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, 'sq_*.csv'))):
    for row in csv.DictReader(open(f)):
        name = row.get('Kernel_Name', '')
        if 'hope' not in name:
            continue
        m = re.search(r'(k_[a-z_]+)', name)
        short = m.group(1) if m else name[:40]
        acc[short][row['Counter_Name']].append(float(row['Counter_Value']))
for k, cs in acc.items():
    print(f'== {k}')
    vals = {}
    for c, v in cs.items():
        half = v[len(v) // 2:]
        vals[c] = sum(half) / max(len(half), 1)
        print(f'   {c:28s} {vals[c]:16.1f}   ({len(v)} launches)')
    w = vals.get('SQ_WAVES')
    if w:
        for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_SMEM', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR'):
            if c in vals:
                print(f'   per wave {c:20s} {vals[c] / w:12.1f}')
    wc = vals.get('SQ_WAVE_CYCLES')
    if wc:
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_SCA'):
            if c in vals:
                print(f'   frac of wave cycles {c:20s} {vals[c] / wc:8.3f}')
