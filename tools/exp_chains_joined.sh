run() { tag=$1; shift; env "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']], sum(v for k,v in d['parity_check'].items() if 'mismatch' in k))"; }
for M in normal dlp; do for NS in 8192 16384 32768 65536; do
A="--rs-join joined --mix $M --scenes $NS"
run "one-chain $A" HOPE_CHAINS=1 timeout 300 python bench.py $A --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
run "two       $A" HOPE_CHAINS=2 HOPE_PIPE_CALLER=0 timeout 300 python bench.py $A --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
done; done
for M in normal dlp; do for NS in 16384 65536; do
A="--image --mix $M --scenes $NS"
run "one-chain $A" HOPE_CHAINS=1 timeout 300 python bench.py $A --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
run "two       $A" HOPE_CHAINS=2 HOPE_PIPE_CALLER=0 timeout 300 python bench.py $A --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
done; done
