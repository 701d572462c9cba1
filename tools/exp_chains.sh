# single-class workloads (every scene in one obstacle-tile class): one chain on the caller's stream (HOPE_CHAINS=1) vs the library's
# default (automatic: two sub-chains of the class in the pipelined two-stream form inside the measured ranges, hope_env_step)
run() { tag=$1; shift; env "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$tag', round(d['ms_per_step'],4), [round(x,4) for x in d['repeat']['ms_per_step']], sum(v for k,v in d['parity_check'].items() if 'mismatch' in k))"; }
for M in normal dlp; do for NS in 4096 8192 16384 32768 65536 131072; do
A="--mix $M --scenes $NS"
run "one-chain $A" HOPE_CHAINS=1 timeout 300 python bench.py $A --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
run "default   $A" timeout 300 python bench.py $A --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
[ $M = dlp ] && [ $NS -le 8192 ] && run "forced-2  $A" HOPE_AUTO_CHAINS=1:1073741824:1:1073741824 timeout 300 python bench.py $A --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
done; done
run "default mixed 65536" timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
run "image normal 65536 one-chain" HOPE_CHAINS=1 timeout 300 python bench.py --image --mix normal --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
run "image normal 65536 HOPE_CHAINS=2" HOPE_CHAINS=2 HOPE_PIPE_CALLER=0 timeout 300 python bench.py --image --mix normal --steps 30 --warmup 10 --no-cpu-baseline --repeat-passes 2 2>/dev/null
