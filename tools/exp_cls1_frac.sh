run() { tag=$1; shift; env "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']], sum(v for k,v in d['parity_check'].items() if 'mismatch' in k))"; }
for i in 1 2; do
for F in 0 0.30 0.36 0.42 0.50; do
run "frac $F 65536" HOPE_CLS1_FRAC=$F timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
done; done
for F in 0 0.30 0.36 0.42 0.50; do
run "frac $F 32768" HOPE_CLS1_FRAC=$F timeout 300 python bench.py --scenes 32768 --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
run "frac $F 131072" HOPE_CLS1_FRAC=$F timeout 300 python bench.py --scenes 131072 --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
done
