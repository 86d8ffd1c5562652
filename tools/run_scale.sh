#!/bin/bash
# The exact multi-GPU lines for BASELINE.json's configurations (one process per GPU, RCCL over xGMI for the final
# gather only; every rank samples its own whole batches, so scaling is weak).  Run on a node with N MI355X.
#   tools/run_scale.sh 8            # configs[1] at 1/2/4/8 GPUs, then configs[3] and configs[4] at 8
set -euo pipefail
cd "$(dirname "$0")/.."
N=${1:-8}
PORT=${MASTER_PORT:-29511}
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() {   # run <gpus> <bench args...>
  local n=$1; shift
  if [[ $n == 1 ]]; then python bench.py --gpus 1 "$@"
  else python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$PORT" bench.py --gpus "$n" "$@"
  fi
}
# configs[1]: batch 256 per GPU, 1000 steps (the headline metric at 1/2/4/8 GPUs)
for n in 1 2 4 8; do [[ $n -le $N ]] && run $n --steps 1000 --warmup 20; done
# configs[3]: batch 8192 sharded 1024 per GPU
run "$N" --batch 1024 --steps 1000 --warmup 20 --cpu-steps 0
# configs[4]: large-molecule stress, 40-80 atoms, k = 32, batch 4096 = 512 per GPU
run "$N" --batch 512 --atoms 40,80 --knn 32 --steps 1000 --warmup 20 --cpu-steps 0 --concurrent 0
