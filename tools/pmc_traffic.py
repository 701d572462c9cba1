#!/usr/bin/env python3
"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE -- they cannot share a pass on gfx950: TCC has 4
slots, FETCH_SIZE takes 3 and WRITE_SIZE 2) into HBM bytes per launch for each hope kernel.

Units / corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly 1/2 of the bytes of a wide coalesced read, so it is doubled.  WRITE_SIZE and narrow reads are uncalibrated
by the guide; the numbers are therefore an estimate of the memory-side traffic, good for ratios.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter or 'hope' not in r['Kernel_Name']:
            continue
        name = r['Kernel_Name']
        for k in ('k_kinematics', 'k_env_step', 'k_obs_pair', 'k_motion_pair', 'k_post', 'k_rs_compact', 'k_rs_words', 'k_rs_segs', 'k_rs_validate', 'k_bev_image', 'k_bev_prep'):
            if k in name:
                acc[k].append(float(r['Counter_Value']))
                if k in ('k_obs_pair', 'k_motion_pair'):                      # the small-tile class's observation launch (two scenes per wave): a launch of the
                    acc['k_env_step'].append(float(r['Counter_Value']))   # step kernel for bench.py's per-launch average (4 launches per step)
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {'unit': 'bytes per launch (average over all launches of the kernel in the profiled run)',
           'correction': 'FETCH_SIZE x2 (gfx950 wide-read under-count), KiB -> bytes', 'kernels': {}}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, (0.0, 0))
        w, nw = write.get(k, (0.0, 0))
        out['kernels'][k] = {'fetch_bytes': 2 * f * 1024, 'write_bytes': w * 1024, 'hbm_bytes': (2 * f + w) * 1024,
                             'launches_fetch_pass': nf, 'launches_write_pass': nw}
    dom = 'k_env_step'
    out['hbm_bytes_per_launch'] = out['kernels'].get(dom, {}).get('hbm_bytes')
    out['kernel'] = dom
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
