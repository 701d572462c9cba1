#!/usr/bin/env python3
"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this library's access widths (the guide's x2 holds for
16-byte-per-lane streaming reads only).  hope_debug_traffic sweeps a 2 GiB buffer (8 x the Infinity Cache) once per mode with a
known byte count.
  cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/cf -- python $R/tools/pmc_calib.py
            rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/cw -- python $R/tools/pmc_calib.py
  python tools/pmc_calib.py --reduce /tmp/cf /tmp/cw > profiles/rNN_pmc_calibration.json"""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BYTES = 2 << 30
MODES = {0: 'read 16 B/lane', 1: 'read 8 B/lane', 2: 'read 4 B/lane', 3: 'read one 8 B word per 64 B line',
         4: 'write 16 B/lane', 5: 'write 8 B/lane', 6: 'write 4 B/lane', 7: 'write one 8 B word per 64 B line'}


def reduce(df, dw):
    out = {'buffer_bytes': BYTES, 'unit': 'counter KiB x 1024 per launch; factor = bytes the sweep requested / counter bytes', 'modes': {}}
    for counter, d in (('FETCH_SIZE', df), ('WRITE_SIZE', dw)):
        f = sorted(glob.glob(d + '/**/*counter_collection.csv', recursive=True))[0]
        acc = {}
        for r in csv.DictReader(open(f)):
            m = re.search(r'k_traffic_calib<(\d)>', r['Kernel_Name'])
            if m and r['Counter_Name'] == counter:
                acc.setdefault(int(m.group(1)), []).append(float(r['Counter_Value']) * 1024)
        for mode, v in sorted(acc.items()):
            v = v[1:] or v                                     # (the first sweep of a mode warms the TLB)
            mean = sum(v) / len(v)
            requested = BYTES if mode not in (3, 7) else BYTES // 8
            e = out['modes'].setdefault(MODES[mode], {'requested_bytes': requested, 'lines_touched_bytes': BYTES})
            e[counter + '_bytes'] = mean
            relevant = (counter == 'FETCH_SIZE') == (mode < 4)
            if relevant and mean > 0:
                e['factor_requested'] = requested / mean
                e['factor_lines'] = BYTES / mean
    print(json.dumps(out, indent=1))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--reduce':
        return reduce(sys.argv[2], sys.argv[3])
    import ctypes as C
    import torch
    from hope_amd import _lib as L
    lib = L.load_library()
    buf = torch.zeros(BYTES // 8, dtype=torch.float64, device='cuda')
    for mode in range(8):
        for _ in range(3):
            L.check(lib.hope_debug_traffic(mode, BYTES, C.c_void_p(buf.data_ptr()), None), 'hope_debug_traffic')
            torch.cuda.synchronize()


if __name__ == '__main__':
    main()
