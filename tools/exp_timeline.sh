mkdir -p gpurun_out; R=$PWD; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl2
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl2 -- python $R/bench.py --steps 8 --warmup 8 --preroll 100 --no-cpu-baseline --witness 0 --repeat-passes 0 > /dev/null 2>&1
python - <<PY > $O/tl_pipe.txt
import csv, glob, re
f = sorted(glob.glob('/tmp/tl2/**/*kernel_trace.csv', recursive=True))[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_' in r['Kernel_Name'] and ('hope' in r['Kernel_Name'] or 'k_rs_compact' in r['Kernel_Name'])]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-60:]
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    m = re.search(r'(k_[a-z_]+)(<[^>]*>)?', r['Kernel_Name'])
    name = m.group(1) + (m.group(2) or '')
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    print(f"q{r.get('Queue_Id','?'):>3} {s:8.1f} -> {e:8.1f} ({e-s:6.1f}) grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>8} {name}")
PY
