# joined form (bench.py --rs-join joined): roles 1, 3, 5 kept as in the default (they decide the deferred form), the others permuted
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --rs-join joined --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 --witness 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']])"; }
run "default(1,2,3,7,5,6,4)" A=1
python - <<'PY' > /tmp/perms.txt
import itertools
for b,d,f,g in itertools.permutations([2,4,6,7],4):
    print(','.join(map(str,[1,b,3,d,5,f,g])))
PY
while read P; do run "$P" HOPE_SIDE_PERM=$P; done < /tmp/perms.txt
