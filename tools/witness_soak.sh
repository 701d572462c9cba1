# bench.py's oracle witness over many seeds with the final code: the timed configuration (mixed, 65 536 scenes) and the single-class
# workloads whose deferred steps run as two sub-chains.  One line per run: seed / workload, the parity_check object.
w() { tag=$1; shift; timeout 600 python bench.py "$@" --steps 250 --warmup 5 --witness 8192 --no-cpu-baseline --repeat-passes 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); p=d['parity_check']; p.pop('checker',None); print('$tag', json.dumps(p))"; }
for s in 1 2 3 4 5 6 7 8; do w "seed $s mixed 65536" --seed $s; done
for s in 1 2 3; do w "seed $s generated lots 65536 (two sub-chains)" --seed $s --mix normal; done
for s in 1 2 3; do w "seed $s Dragon-Lake 16384 (two sub-chains)" --seed $s --mix dlp --scenes 16384; done
