#!/bin/bash
# Vector / scalar instructions and wave-cycles of the small-tile class's MOTION launch per scene: two scenes per wave (k_motion_pair) against
# the one-scene kernel (HOPE_MOTION_PAIR=0), same bench workload (steady state).  Run through gpurun from the repo root:
#   bash tools/motion_probe.sh > gpurun_out/motion_insts.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
for MP in 1 0; do
  rm -rf /tmp/mp_$MP
  HOPE_MOTION_PAIR=$MP timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d /tmp/mp_$MP -- python $R/bench.py --steps 4 --warmup 4 --preroll 100 --refresh-every 0 --no-cpu-baseline --witness 0 --repeat-passes 0 > /dev/null 2>&1
  python - $MP <<'P'
import csv, glob, sys, collections
mp = sys.argv[1]
f = sorted(glob.glob(f'/tmp/mp_{mp}/**/*counter_collection.csv', recursive=True))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'k_motion_pair' in n: key = 'k_motion_pair (49 152 scenes, 2 per wave)'
    elif 'k_env_step' in n and ', 1, ' in n:
        g = int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0)
        key = f'k_env_step<motion> grid {g} ({g // 64} scenes, 1 per wave)'
    else: continue
    acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, cs in sorted(acc.items()):
    scenes = 49152 if 'pair' in key else int(key.split('(')[1].split()[0])
    print(f'HOPE_MOTION_PAIR={mp}  {key:60s} ' + '  '.join(f'{c} {sum(v[len(v)//2:]) / len(v[len(v)//2:]) / scenes:8.1f}' for c, v in sorted(cs.items())) + '  per scene')
P
done
