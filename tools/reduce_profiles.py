#!/usr/bin/env python3
"""Reduces the rocprofv3 output directories written by tools/collect_profiles.sh to the small files kept under
profiles/: kernel statistics (csv + top lines), per-dispatch PMC values of the hope kernels, and the per-launch
memory-side traffic (tools/pmc_traffic.py)."""
import csv
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
KERNELS = ('k_kinematics', 'k_env_step', 'k_obs_pair', 'k_motion_pair', 'k_post', 'k_rs_compact', 'k_rs_words', 'k_rs_segs', 'k_rs_validate', 'k_bev_image', 'k_bev_prep', 'k_bev_static')


def find(d, pat):
    hits = glob.glob(os.path.join(d, '**', pat), recursive=True)
    return hits[0] if hits else None


def kernel_stats(src_dir, out_csv, out_top):
    f = find(src_dir, '*kernel_stats.csv')
    if not f:
        return
    rows = list(csv.DictReader(open(f)))
    with open(out_csv, 'w') as g:
        w = csv.DictWriter(g, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_NONNUMERIC)
        w.writeheader()
        for r in rows[:24]:
            w.writerow(r)
    with open(out_top, 'w') as g:
        for r in rows[:10]:
            g.write(f"{r['Name'][:70]:70s} calls={int(r['Calls']):4d} avg_ns={float(r['AverageNs']):12.0f} "
                    f"max_ns={int(float(r['MaxNs'])):9d} pct={r['Percentage']}\n")
        # the launches bench.py's roofline averages over: every k_env_step instantiation + k_obs_pair (the small-tile class's observation launch)
        st = [r for r in rows if 'k_env_step' in r['Name'] or 'k_obs_pair' in r['Name'] or 'k_motion_pair' in r['Name']]
        if st:
            calls = sum(int(r['Calls']) for r in st)
            tot = sum(float(r['TotalDurationNs']) for r in st) if 'TotalDurationNs' in st[0] else sum(float(r['AverageNs']) * int(r['Calls']) for r in st)
            g.write(f"{'step kernel, all launches (k_env_step<..> + k_motion_pair + k_obs_pair)':70s} calls={calls:4d} avg_ns={tot / calls:12.0f}\n")


def pmc(src_dir, counter, out_csv):
    f = find(src_dir, '*counter_collection.csv')
    if not f:
        return None
    with open(out_csv, 'w') as g:
        g.write('Dispatch_Id,Kernel,Grid_Size,LDS_Block_Size,VGPR_Count,Counter_Name,Counter_Value_KiB\n')
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            k = next((k for k in KERNELS if k in r['Kernel_Name']), None)
            if k:
                g.write(f"{r['Dispatch_Id']},{k},{r['Grid_Size']},{r.get('LDS_Block_Size', '')},{r.get('VGPR_Count', '')},"
                        f"{counter},{float(r['Counter_Value']):.6f}\n")
    return f


def busy(o, v, out_txt):
    """derived metrics (percent), averaged over the launches of the second half of the run (steady state)"""
    import collections
    acc = collections.defaultdict(list)
    for d in sorted(glob.glob(os.path.join(o, f'busy{v}_[0-9]'))):
        f = find(d, '*counter_collection.csv')
        if not f:
            continue
        rows = list(csv.DictReader(open(f)))
        for r in rows[len(rows) // 2:]:
            k = next((k for k in KERNELS if k in r['Kernel_Name']), None)
            if k:
                acc[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
    if not acc:
        return
    names = sorted({c for _, c in acc})
    with open(out_txt, 'w') as g:
        g.write(f"{'kernel':16s}" + ''.join(f'{n:>18s}' for n in names) + '   (percent, rocprofv3 derived metrics)\n')
        for k in KERNELS:
            if any((k, n) in acc for n in names):
                g.write(f'{k:16s}' + ''.join(f"{sum(acc[(k, n)]) / max(len(acc[(k, n)]), 1):18.2f}" for n in names) + '\n')


def main():
    o, tag = sys.argv[1], sys.argv[2]
    busy(o, '', os.path.join(o, f'{tag}_pmc_busy.txt'))
    busy(o, '_image', os.path.join(o, f'{tag}_pmc_busy_image.txt'))
    kernel_stats(os.path.join(o, 'kstats'), os.path.join(o, f'{tag}_kernel_stats.csv'), os.path.join(o, f'{tag}_kernel_stats_top.txt'))
    kernel_stats(os.path.join(o, 'kstats_img'), os.path.join(o, f'{tag}_kernel_stats_image.csv'),
                 os.path.join(o, f'{tag}_kernel_stats_image_top.txt'))
    for v in ('', '_image'):
        raw = {}
        for c in ('FETCH_SIZE', 'WRITE_SIZE'):
            raw[c] = pmc(os.path.join(o, f'pmc{v}_{c}'), c, os.path.join(o, f'{tag}_pmc{v}_{c.lower()}.csv'))
        if raw['FETCH_SIZE'] and raw['WRITE_SIZE']:
            subprocess.check_call([sys.executable, os.path.join(HERE, 'pmc_traffic.py'), raw['FETCH_SIZE'], raw['WRITE_SIZE'],
                                   os.path.join(o, f'{tag}_pmc_traffic{v}.json')], stdout=subprocess.DEVNULL)


if __name__ == '__main__':
    main()
