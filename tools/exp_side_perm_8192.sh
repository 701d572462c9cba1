# stream roles under the final (pipelined) launch structure: HOPE_SIDE_PERM = stream index (creation order) of roles 1..7; in the default
# run (no image, class 0 env chain on the caller's stream) roles 1 (search stream of the small-tile class), 3 (env stream of the
# large-tile class) and 5 (search stream of the large-tile class) are in use
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --scenes 8192 --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 --witness 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']])"; }
run "default(1,2,3,7,5,6,4)" A=1
python - <<'PY' > /tmp/perms.txt
import itertools, random
random.seed(3)
seen=set()
out=[]
# roles 1,3,5 -> three distinct streams; the rest filled in order
for a,b,c in itertools.permutations(range(1,8),3):
    out.append((a,b,c))
random.shuffle(out)
for a,b,c in out[:30]:
    rest=[x for x in range(1,8) if x not in (a,b,c)]
    perm=[a,rest[0],b,rest[1],c,rest[2],rest[3]]
    print(','.join(map(str,perm)))
PY
while read P; do run "$P" HOPE_SIDE_PERM=$P; done < /tmp/perms.txt
run "default(1,2,3,7,5,6,4)" A=1
