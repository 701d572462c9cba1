#!/bin/bash
O=$PWD/gpurun_out/r05; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/bench.py --image --steps 8 --warmup 8 --preroll 100 --refresh-every 0 --no-cpu-baseline --witness 0 --repeat-passes 0 > /dev/null 2>&1
python - <<'P' > $O/timeline_image.txt
import csv, glob, re
f = sorted(glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True))[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'k_' in r['Kernel_Name'] or 'copy' in r['Kernel_Name'].lower()]
rows = rows[-75:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    m = re.search(r'(k_[a-z_]+)(<[^>]*>)?', r['Kernel_Name'])
    name = (m.group(1) + (m.group(2) or '')) if m else r['Kernel_Name'][:40]
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    print(f"q{r.get('Queue_Id','?'):>3} {s:8.1f} -> {e:8.1f} ({e-s:6.1f}) grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>8} {name}")
P
cat $O/timeline_image.txt
