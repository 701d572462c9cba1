# stream roles for a single-class batch in the two-sub-chain form (roles 1, 3, 4, 5 in use; the caller's stream only joins)
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --mix normal --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 --witness 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']])"; }
run "default(1,2,3,7,5,6,4)" A=1
python - <<'PY' > /tmp/perms.txt
import itertools, random
random.seed(5)
out=list(itertools.permutations(range(1,8),4))
random.shuffle(out)
for a,b,c,d in out[:30]:
    rest=[x for x in range(1,8) if x not in (a,b,c,d)]
    perm=[a,rest[0],b,c,d,rest[1],rest[2]]
    print(','.join(map(str,perm)))
PY
while read P; do run "$P" HOPE_SIDE_PERM=$P; done < /tmp/perms.txt
