#!/usr/bin/env python3
"""Builds and runs tools/valu_rate.hip on the GPU and writes the measured issue rates (wave-instructions per second per
instruction class, at 1 / 2 / 4 / 8 waves per SIMD) as JSON:

    gpurun -- 'python tools/valu_rate.py gpurun_out/r03_valu_rates.json'

`summary` holds, per class, the best rate over the occupancies (the ceiling bench.py's mix-weighted VALU-issue roofline
uses: profiles/r03_valu_rates.json) and the implied cycles per wave64 instruction per SIMD at the measured clock."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'valu_rates.json')
    exe = '/tmp/valu_rate.bin'
    hipcc = '/opt/rocm/bin/hipcc'
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', os.path.join(ROOT, 'tools', 'valu_rate.hip'), '-o', exe])
    txt = subprocess.check_output([exe] + sys.argv[2:3]).decode()
    data = json.loads(txt)
    n_simd = data['cus'] * 4
    summary = {}
    for r in data['results']:
        s = summary.setdefault(r['inst'], {'class': r['class'], 'wave_insts_per_s': 0.0})
        if r['wave_insts_per_s'] > s['wave_insts_per_s']:
            s.update(wave_insts_per_s=r['wave_insts_per_s'], at_waves_per_simd=r['waves_per_simd'], memtime_GHz=r['memtime_GHz'],
                     cycles_per_inst_at_2p4GHz=n_simd * 2.4e9 / r['wave_insts_per_s'])
        s.setdefault('by_occupancy', {})[str(r['waves_per_simd'])] = r['wave_insts_per_s']
    data['summary'] = summary
    # the class ceilings bench.py uses: float64 arithmetic (add / mul / fma), float64 transcendental seeds, everything else
    f64 = min(summary[k]['wave_insts_per_s'] for k in ('v_fma_f64', 'v_add_f64', 'v_mul_f64'))
    trans = min(summary[k]['wave_insts_per_s'] for k in ('v_rcp_f64', 'v_sqrt_f64'))
    b32 = min(summary[k]['wave_insts_per_s'] for k in ('v_fma_f32', 'v_add_u32', 'v_and_b32', 'v_mov_b32'))
    data['ceilings'] = {'f64_arith_wave_insts_per_s': f64, 'f64_trans_wave_insts_per_s': trans, 'other_valu_wave_insts_per_s': b32,
                        'note': 'best rate over 1/2/4/8 waves per SIMD; "other" = the slowest of v_fma_f32, v_add_u32, v_and_b32, v_mov_b32 '
                                '(the 32-bit ALU classes; DPP moves, f64 min / compare and readlane are listed separately in summary)'}
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    json.dump(data, open(out, 'w'), indent=1)
    for k, v in summary.items():
        print(f"{k:32s} {v['wave_insts_per_s']:.3e} wave-inst/s  {v['cycles_per_inst_at_2p4GHz']:.2f} cyc/inst/SIMD @2.4GHz  (best at {v['at_waves_per_simd']} waves/SIMD)")
    print(json.dumps(data['ceilings']))


if __name__ == '__main__':
    main()
