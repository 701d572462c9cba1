#!/bin/bash
# 1 / 2 / 4 / 8-GPU table of bench.py on ONE node: weak scaling (65 536 scenes per GPU) and strong scaling (65 536 scenes in total),
# one rank per GPU over RCCL (the launch line the driver uses).  Prints one row per run; efficiency = value / (N x value at N = 1).
#   bash tools/run_scale.sh [max_gpus] [extra bench.py args]
set -u
MAXG=${1:-8}; shift || true
R=$(cd "$(dirname "$0")/.." && pwd)
NG=$(python -c "import torch; print(torch.cuda.device_count())")
[ "$NG" -lt "$MAXG" ] && MAXG=$NG
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {  # n_gpus scaling
  if [ "$1" -eq 1 ]; then python $R/bench.py --gpus 1 --scaling $2 --steps 40 --warmup 10 --no-cpu-baseline --witness 0 --repeat-passes 0 "${@:3}" 2>/dev/null | tail -1
  else python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((29500 + $1)) $R/bench.py --gpus $1 --scaling $2 --steps 40 --warmup 10 --no-cpu-baseline --witness 0 --repeat-passes 0 "${@:3}" 2>/dev/null | tail -1
  fi
}
for MODE in weak strong; do
  BASE=""
  for N in 1 2 4 8; do
    [ "$N" -gt "$MAXG" ] && break
    LINE=$(run $N $MODE "$@")
    python - "$MODE" "$N" "$BASE" <<PY
import json, sys
mode, n, base = sys.argv[1], int(sys.argv[2]), sys.argv[3]
d = json.loads('''$LINE''')
v = d['value']
eff = v / (n * float(base)) if base and mode == 'weak' else (v / float(base) / n if base else 1.0)
print(f"{mode:6s} {n} GPU(s): {v / 1e6:8.2f} M env-steps/s  {d['ms_per_step']:.3f} ms/step  scenes/GPU {d['config']['scenes_per_gpu']}  efficiency {eff:.3f}")
PY
    [ -z "$BASE" ] && BASE=$(python -c "import json; print(json.loads('''$LINE''')['value'])")
  done
done
