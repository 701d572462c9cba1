# A/B of two builds on the same box, interleaved: hope_amd/libhope_env.so vs hope_amd/libhope_env_b.so
run() { tag=$1; shift; env "$@" python bench.py --steps 30 --warmup 10 --no-cpu-baseline --witness 0 --repeat-passes 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']])"; }
for i in 1 2 3; do
run A A=1
run B HOPE_AMD_LIB=$PWD/hope_amd/libhope_env_b.so
done
