# stream roles for the step with the image (bench.py --image): all seven library roles may matter
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --image --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 --witness 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']])"; }
run "default(1,2,3,7,5,6,4)" A=1
python - <<'PY' > /tmp/perms.txt
import itertools, random
random.seed(11)
out=list(itertools.permutations(range(1,8),7))
random.shuffle(out)
for p in out[:40]:
    print(','.join(map(str,p)))
PY
while read P; do run "$P" HOPE_SIDE_PERM=$P; done < /tmp/perms.txt
run "default(1,2,3,7,5,6,4)" A=1
