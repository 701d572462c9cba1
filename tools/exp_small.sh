run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --witness 0 --repeat-passes 3 $EXTRA 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']])"; }
for NS in 4096 8192; do
EXTRA="--scenes $NS" run d_$NS A=1
EXTRA="--scenes $NS" run postsearch_$NS HOPE_POST_SEARCH=1
done
for NS in 16384 32768; do
EXTRA="--scenes $NS" run d_$NS A=1
EXTRA="--scenes $NS" run nopostsearch_$NS HOPE_POST_SEARCH=0
EXTRA="--scenes $NS" run onelaunch_nopostsearch_$NS HOPE_POST_SEARCH=0 HOPE_SPLIT_MIN=65536
done
