#!/bin/bash
# the batch-size / mode sweep of tools/collect_profiles.sh alone (re-run after a change that only moves these lines)
TAG=${1:-r05}; R=$PWD; O=$R/gpurun_out/final; mkdir -p $O
T="timeout 400"
py() { $T python "$@" 2>>$O/stderr.log; }
{
  for NS in 4096 8192 16384 32768 65536 131072; do echo "== --scenes $NS"; py $R/bench.py --scenes $NS --no-cpu-baseline --witness 0 --repeat-passes 2 --steps 40 --warmup 10 | tail -1; done
  for NS in 4096 8192 16384; do echo "== --scenes $NS HOPE_RS_DEBUG=0x20000 (no screen pass)"; HOPE_RS_DEBUG=0x20000 $T python $R/bench.py --scenes $NS --no-cpu-baseline --witness 0 --repeat-passes 2 --steps 40 --warmup 10 2>/dev/null | tail -1; done
  echo "== config 2: --stages motion --scenes 4096"; py $R/bench.py --scenes 4096 --stages motion --no-cpu-baseline --steps 40 --warmup 10 --repeat-passes 0 --witness 0 | tail -1
  echo "== config 3: --scenes 16384 (full step)"; py $R/bench.py --scenes 16384 --no-cpu-baseline --steps 40 --warmup 10 --repeat-passes 2 | tail -1
  echo "== the driver's exact form: --steps 20 --warmup 5"; py $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline | tail -1
  echo "== the driver's exact form, second run"; py $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline | tail -1
  echo "== the driver's exact form, third run"; py $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline | tail -1
} > $O/${TAG}_bench_modes_2.txt
python - <<P
import json
lab=None
for l in open('$O/${TAG}_bench_modes_2.txt'):
    l=l.rstrip()
    if l.startswith('=='): lab=l; continue
    if l.startswith('{'):
        d=json.loads(l); r=d.get('repeat') or {}
        print(lab, '|', round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), 'ms | steady', r.get('median') and round(r['median'],4), '| joined', r.get('joined_ms_per_step') and round(r['joined_ms_per_step'],4), '| refresh', (d.get('pool_refresh') or {}).get('every_steps'), (d.get('pool_refresh') or {}).get('refresher_commits'))
P
