# A/B of two builds on the same box, interleaved: hope_amd/libhope_env.so vs hope_amd/libhope_env_b.so (HOPE_BUILD_DEFS=... HOPE_AMD_LIB=... build)
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']], sum(v for k,v in d['parity_check'].items() if 'mismatch' in k))"; }
for i in 1 2 3; do
run A A=1
run B HOPE_AMD_LIB=$PWD/hope_amd/libhope_env_b.so
done
run wpc1_6 HOPE_RS_WPC1=6
run wpc1_10 HOPE_RS_WPC1=10
run wpc1_13 HOPE_RS_WPC1=13
run wpc0_20 HOPE_RS_WPC0=20
run rsprio0 HOPE_RS_PRIO=0
