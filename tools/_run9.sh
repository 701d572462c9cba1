cd $GRAFT_REPO_ROOT
for NS in 4096 8192 16384 24576 32768; do for V in "A=1" "HOPE_SPLIT_MIN=1"; do echo -n "== $NS $V: "; env $V timeout 300 python bench.py --scenes $NS --no-cpu-baseline --witness 0 --repeat-passes 3 --steps 40 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']])"; done; done
