# A = default block -> scene map (lines kept on one XCD), B = hope_amd/libhope_env_b1.so built with -DHOPE_XCD_XOR; other scene mixes and sizes
run() { tag=$1; shift; env "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']], sum(v for k,v in d['parity_check'].items() if 'mismatch' in k))"; }
for i in 1 2; do
for A in "--mix normal" "--mix dlp --scenes 16384" "--scenes 8192" "--scenes 131072" "--image"; do
run "A $A" timeout 300 python bench.py $A --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 3 2>/dev/null
run "B $A" HOPE_AMD_LIB=$PWD/hope_amd/libhope_env_b1.so timeout 300 python bench.py $A --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 3 2>/dev/null
done
done
