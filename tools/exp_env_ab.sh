#!/bin/bash
# A/B of run-time variants on the same box, interleaved:  bash tools/exp_env_ab.sh ROUNDS "tag1:ENV=V,ENV2=V" "tag2:" ...
# (a variant may also name another build of the library: HOPE_AMD_LIB=$PWD/hope_amd/libhope_env_b1.so); BENCH_ARGS adds bench.py flags
R=${1:-3}; shift
run() { tag=$1; shift; env "$@" timeout 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 3 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', 'driver-form', round(d['ms_per_step'],4), 'steady', [round(x,4) for x in d['repeat']['ms_per_step']], 'mismatch', sum(v for k,v in (d['parity_check'] or {}).items() if 'mismatch' in k))"; }
for i in $(seq $R); do
  for v in "$@"; do
    tag=${v%%:*}; envs=${v#*:}
    if [ -z "$envs" ]; then run $tag A=1; else run $tag $(echo $envs | tr ',' ' '); fi
  done
done
