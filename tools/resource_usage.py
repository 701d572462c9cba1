#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS usage of libhope_env's device code, from hipcc -Rpass-analysis=kernel-resource-usage
(what the review quotes: VGPRs, SGPR / VGPR spills, scratch bytes per lane, occupancy).  Cross-compiles, no GPU needed.
    python tools/resource_usage.py [--all] > profiles/rNN_kernel_resource_usage.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hope_amd.build import FLAGS  # noqa: E402

CS = os.path.join(ROOT, 'hope_amd', 'csrc')
KEYS = ['VGPRs', 'AGPRs', 'ScratchSize [bytes/lane]', 'SGPRs', 'VGPRs Spill', 'SGPRs Spill', 'Occupancy [waves/SIMD]', 'LDS Size [bytes/block]']


def demangle(names):
    out = subprocess.run(['c++filt'] + names, capture_output=True, text=True).stdout.split('\n')
    return [o.replace('hope::', '').replace('(anonymous namespace)::', '') for o in out[:len(names)]]


def main():
    show_all = '--all' in sys.argv
    rows = []
    for src in ('hope_env.hip', 'hope_rs.hip', 'hope_bev.hip'):
        flags = [f for f in FLAGS if f not in ('-shared', '-pthread')] + os.environ.get('HOPE_BUILD_DEFS', '').split()
        cmd = ['/opt/rocm/bin/hipcc'] + flags + ['-I' + os.path.join(ROOT, 'include'), '-I' + CS, '--cuda-device-only',
               '-Rpass-analysis=kernel-resource-usage', '-c', os.path.join(CS, src), '-o', '/dev/null']
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        cur = None
        for line in err.split('\n'):
            m = re.search(r'remark: (?:Function )?Name: (\S+)', line)
            if m:
                cur = {'name': m.group(1), 'src': src}
                rows.append(cur)
                continue
            m = re.search(r'remark:\s+(.*?): (\d+)', line)
            if m and cur is not None:
                cur[m.group(1).strip()] = int(m.group(2))
    names = demangle([r['name'] for r in rows])
    print(f'{"kernel":78s} {"VGPR":>5s} {"SGPR":>5s} {"scratch B":>9s} {"VGPR spill":>10s} {"SGPR spill":>10s} {"waves/SIMD":>10s} {"LDS B":>7s}')
    for r, n in zip(rows, names):
        n = re.sub(r'\(.*', '', n).replace('void ', '')
        if not show_all and not n.startswith('k_'):
            continue
        print(f'{n[:78]:78s} {r.get("VGPRs", 0):5d} {r.get("SGPRs", r.get("TotalSGPRs", 0)):5d} {r.get("ScratchSize [bytes/lane]", 0):9d} '
              f'{r.get("VGPRs Spill", r.get("VGPR Spill", 0)):10d} {r.get("SGPRs Spill", r.get("SGPR Spill", 0)):10d} '
              f'{r.get("Occupancy [waves/SIMD]", 0):10d} {r.get("LDS Size [bytes/block]", 0):7d}')


if __name__ == '__main__':
    main()
