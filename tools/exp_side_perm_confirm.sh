run() { tag=$1; shift; env "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']])"; }
for i in 1 2; do
for P in 1,2,3,7,5,6,4 1,3,7,6,4,5,2 6,2,3,1,4,5,7 4,5,7,3,2,6,1 4,5,6,3,1,2,7 6,2,7,1,3,5,4; do
run "image    $P" HOPE_SIDE_PERM=$P timeout 300 python bench.py --image --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 --witness 0 2>/dev/null
run "no-image $P" HOPE_SIDE_PERM=$P timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 --witness 0 2>/dev/null
done; done
for P in 1,2,3,7,5,6,4 1,3,7,6,4,5,2 6,2,3,1,4,5,7 4,5,7,3,2,6,1 4,5,6,3,1,2,7 6,2,7,1,3,5,4; do
run "image 8192    $P" HOPE_SIDE_PERM=$P timeout 300 python bench.py --image --scenes 8192 --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 --witness 0 2>/dev/null
run "no-image 8192 $P" HOPE_SIDE_PERM=$P timeout 300 python bench.py --scenes 8192 --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 --witness 0 2>/dev/null
run "normal 65536  $P" HOPE_SIDE_PERM=$P timeout 300 python bench.py --mix normal --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 --witness 0 2>/dev/null
done
