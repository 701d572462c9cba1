#!/usr/bin/env python3
"""Timeline of one bench step from a rocprofv3 kernel trace: start / end of every kernel launch relative to the step's first
launch, per stream -- shows what overlaps what and where the launch gaps are.
usage (GPU box):  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $REPO/bench.py --steps 6 --warmup 6 --no-cpu-baseline
                  python tools/timeline.py /tmp/tl [step_index_from_end]"""
import csv
import glob
import re
import sys


def tail(d, n):
    """the last n launches of the trace, as they are (pipelined steps overlap: no step can be cut out)"""
    f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))[0]
    rows = [r for r in csv.DictReader(open(f)) if 'k_' in r['Kernel_Name'] and ('hope' in r['Kernel_Name'] or 'k_rs_compact' in r['Kernel_Name'])]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    rows = rows[-n:]
    t0 = int(rows[0]['Start_Timestamp'])
    for r in rows:
        m = re.search(r'(k_[a-z_]+)(<[^>]*>)?', r['Kernel_Name'])
        name = m.group(1) + (m.group(2) or '')
        s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
        print(f"  q{r.get('Queue_Id', '?'):>3} {s:8.1f} -> {e:8.1f} us  ({e - s:6.1f})  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8}  {name}")


def main():
    d = sys.argv[1]
    if len(sys.argv) > 3 and sys.argv[2] == '--tail':
        return tail(d, int(sys.argv[3]))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))[0]
    rows = [r for r in csv.DictReader(open(f)) if 'hope' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    # a step starts with a k_kinematics launch that follows a k_rs_validate (or the first one)
    starts = [i for i, r in enumerate(rows) if 'k_kinematics' in r['Kernel_Name'] and (i == 0 or 'k_kinematics' not in rows[i - 1]['Kernel_Name'])]
    # group: consecutive kinematics launches belong to the same step
    steps = []
    for i in starts:
        if steps and i - steps[-1] < 4:
            continue
        steps.append(i)
    a = steps[-back - 1]
    b = steps[-back]
    t0 = int(rows[a]['Start_Timestamp'])
    print(f'step of {b - a} launches, {(int(rows[b]["Start_Timestamp"]) - t0) / 1e3:.1f} us to the next step\'s first launch')
    for r in rows[a:b]:
        m = re.search(r'(k_[a-z_]+)(<[^>]*>)?', r['Kernel_Name'])
        name = m.group(1) + (m.group(2) or '') if m else r['Kernel_Name'][:30]
        s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
        q = r.get('Queue_Id', r.get('Stream_Id', '?'))
        print(f'  q{q:>3} {s:8.1f} -> {e:8.1f} us  ({e - s:6.1f})  grid {r.get("Grid_Size_X", r.get("Grid_Size", "?")):>8}  {name}')


if __name__ == '__main__':
    main()
