#!/bin/bash
# scratch: time library variants (SHAPEMOL_LIB=name) with the B = 256 bench, one line per variant
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/ab.txt
for v in "$@"; do
  if [ "$v" == "base" ]; then unset SHAPEMOL_LIB; else export SHAPEMOL_LIB=$v; fi
  timeout -k 10 150 python bench.py --steps 400 --warmup 20 --cpu-steps 0 --concurrent 0 --exact-steps 0 --profile-steps 0 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.log || { echo "$v FAILED" >> gpurun_out/ab.txt; tail -3 gpurun_out/ab_$v.log; continue; }
  python - "$v" >> gpurun_out/ab.txt <<'PY'
import json,sys
l=[x for x in open(f"gpurun_out/ab_{sys.argv[1]}.json") if x.startswith("{")][-1]
d=json.loads(l); print(sys.argv[1], "ms_per_step", round(d["ms_per_step"],4), "value", round(d["value"],1))
PY
done
cat gpurun_out/ab.txt
