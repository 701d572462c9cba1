# A/B of builds on the same box, interleaved: hope_amd/libhope_env.so vs hope_amd/libhope_env_b*.so (HOPE_BUILD_DEFS=... HOPE_AMD_LIB=... build)
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']], sum(v for k,v in d['parity_check'].items() if 'mismatch' in k))"; }
for i in 1 2 3; do
run A A=1
for b in hope_amd/libhope_env_b*.so; do run $(basename $b .so) HOPE_AMD_LIB=$PWD/$b; done
done
