#!/bin/bash
# 1 / 2 / 4 / 8-GPU table of bench.py on ONE node: weak scaling (65 536 scenes per GPU) and strong scaling (65 536 scenes in total),
# one rank per GPU over RCCL (the launch line the driver uses).  Prints one row per run and, last, ONE JSON object with every
# bench line (rank accounting included): {"weak": {"1": {...}, ...}, "strong": {...}}.  efficiency = value / (N x value at N = 1).
#   bash tools/run_scale.sh [max_gpus] [extra bench.py args]            (1-GPU box: HOPE_BENCH_SHARE_GPU=1 FORCE_G=8 puts all
#   ranks on cuda:0 over gloo -- control flow only)
set -u
MAXG=${1:-8}; shift || true
R=$(cd "$(dirname "$0")/.." && pwd)
NG=$(python -c "import torch; print(torch.cuda.device_count())")
[ -n "${FORCE_G:-}" ] && NG=$FORCE_G
[ "$NG" -lt "$MAXG" ] && MAXG=$NG
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$(mktemp -d)
run() {  # n_gpus scaling
  if [ "$1" -eq 1 ]; then python $R/bench.py --gpus 1 --scaling $2 --steps 40 --warmup 10 --no-cpu-baseline --witness 0 --repeat-passes 0 "${@:3}" 2>/dev/null | tail -1
  else python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((29500 + $1)) $R/bench.py --gpus $1 --scaling $2 --steps 40 --warmup 10 --no-cpu-baseline --witness 0 --repeat-passes 0 "${@:3}" 2>/dev/null | tail -1
  fi
}
for MODE in weak strong; do
  for N in 1 2 4 8; do
    [ "$N" -gt "$MAXG" ] && break
    run $N $MODE "$@" > $OUT/${MODE}_$N.json
  done
done
python - "$OUT" <<'PY'
import glob, json, os, sys
out = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], '*.json'))):
    mode, n = os.path.basename(f)[:-5].split('_')
    try:
        out.setdefault(mode, {})[n] = json.loads(open(f).read().strip().split('\n')[-1])
    except Exception as e:
        out.setdefault(mode, {})[n] = {'error': str(e)}
for mode, rows in out.items():
    base = rows.get('1', {}).get('value')
    for n in sorted(rows, key=int):
        d = rows[n]
        if 'value' not in d:
            print(f'{mode:6s} {n} GPU(s): failed ({d.get("error")})'); continue
        eff = d['value'] / (int(n) * base) if base else float('nan')
        rc = d.get('rccl_check') or {}
        print(f"{mode:6s} {n} GPU(s): {d['value'] / 1e6:8.2f} M env-steps/s  {d['ms_per_step']:.3f} ms/step  scenes/GPU {d['config']['scenes_per_gpu']}  "
              f"efficiency {eff:.3f}  ranks {d.get('ranks')}  backend {rc.get('backend')}  all-reduce of ones {rc.get('allreduce_of_ones')}  "
              f"rank ms min/max {d.get('rank_ms_per_step', {}).get('min', 0):.3f}/{d.get('rank_ms_per_step', {}).get('max', 0):.3f}")
print(json.dumps(out))
PY
rm -rf $OUT
