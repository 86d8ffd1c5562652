#!/bin/bash
# scratch: B = 256 bench under library options, one line each:  bash tools/gpu_opt_sweep.sh "lin_waves=8" "lin_waves=12" ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; : > gpurun_out/opt_sweep.txt
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  timeout -k 10 150 python bench.py --steps 400 --warmup 20 --cpu-steps 0 --concurrent 0 --exact-steps 0 --profile-steps 0 $args > gpurun_out/opt.json 2> gpurun_out/opt.log || { echo "$o FAILED" >> gpurun_out/opt_sweep.txt; tail -2 gpurun_out/opt.log; continue; }
  python - "$o" >> gpurun_out/opt_sweep.txt <<'PY'
import json,sys
l=[x for x in open("gpurun_out/opt.json") if x.startswith("{")][-1]
d=json.loads(l); print(sys.argv[1], "ms_per_step", round(d["ms_per_step"],4), "value", round(d["value"],1))
PY
done
cat gpurun_out/opt_sweep.txt
