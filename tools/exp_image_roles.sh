run() { tag=$1; shift; env "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']], sum(v for k,v in d['parity_check'].items() if 'mismatch' in k))"; }
for NS in 4096 8192 16384 32768 65536; do
run "image $NS old-roles" HOPE_SIDE_PERM=1,2,3,7,5,6,4 timeout 300 python bench.py --image --scenes $NS --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
run "image $NS default  " timeout 300 python bench.py --image --scenes $NS --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
done
run "no-image 65536 default" timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
run "image normal 65536 old-roles" HOPE_SIDE_PERM=1,2,3,7,5,6,4 timeout 300 python bench.py --image --mix normal --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
run "image normal 65536 default" timeout 300 python bench.py --image --mix normal --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
echo "== config 4 share: --policy hope --algo rollout --scenes 8192 --image"
HOPE_SIDE_PERM=1,2,3,7,5,6,4 timeout 300 python bench.py --policy hope --algo rollout --scenes 8192 --image --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
timeout 300 python bench.py --policy hope --algo rollout --scenes 8192 --image --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
