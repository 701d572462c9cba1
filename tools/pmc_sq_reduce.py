#!/usr/bin/env python3
"""Per-kernel averages of the raw SQ counters collected by tools/pmc_sq.sh (second half of the launches of each kernel).
Also writes <dir>/sq_counters.json: per-launch averages per kernel and the VALU / SALU instructions of one bench step
(launches per step = launches of the kernel / 16 env steps of the profiled command, rounded)."""
import csv
import json
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
        if short == 'k_rs_validate_f':                     # the float32-filter validation kernel: the launch the library counts as k_rs_validate
            short = 'k_rs_validate'
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

summary = {'unit': 'per launch, average over the second half of the launches in the profiled run', 'kernels': {}}
# tools/pmc_sq.sh profiles `bench.py --steps 4 --warmup 8 --preroll 60`: 60 + 8 + 4 timed + 4 breakdown = 76 env steps with an action
# (plus one action-less reset_obs, which also launches the class kernels once); the count is the second argument
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tot_valu = tot_salu = 0.0
for k, cs in acc.items():
    per = {c: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for c, v in cs.items()}
    nl = len(next(iter(cs.values())))
    per['launches_per_bench_step'] = max(1, round(nl / STEPS))
    summary['kernels'][k] = per
    tot_valu += per.get('SQ_INSTS_VALU', 0.0) * per['launches_per_bench_step']
    tot_salu += per.get('SQ_INSTS_SALU', 0.0) * per['launches_per_bench_step']
summary['valu_insts_per_bench_step'] = tot_valu
summary['salu_insts_per_bench_step'] = tot_salu
json.dump(summary, open(os.path.join(d, 'sq_counters.json'), 'w'), indent=1)
