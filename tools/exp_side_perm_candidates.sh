# the library's default stream roles (third session: 1,6,3,2,5,4,7 + role 7 for the second sub-chain) against the former default
# (HOPE_SIDE_PERM=1,2,3,7,5,6,4; with it the second sub-chain's env stream, role 7, is the 4th stream) over the launch forms
run() { tag=$1; shift; env "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']], sum(v for k,v in d['parity_check'].items() if 'mismatch' in k))"; }
for A in "" "--scenes 8192" "--scenes 16384" "--scenes 32768" "--scenes 131072" "--rs-join joined" "--rs-join joined --scenes 32768" "--rs-join joined --scenes 16384" "--rs-join joined --scenes 8192" "--mix normal" "--mix normal --scenes 32768" "--mix dlp --scenes 16384" "--mix dlp --scenes 8192" "--rs-join joined --mix normal --scenes 16384" "--rs-join joined --mix dlp --scenes 16384" "--image" "--image --scenes 8192"; do
run "default | $A" timeout 300 python bench.py $A --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
done
