#!/bin/bash
# scratch: per-kernel average durations of library variants (rocprofv3 --kernel-trace --stats, B = 256, 100 steps)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/abp.txt
for v in "$@"; do
  if [ "$v" == "base" ]; then unset SHAPEMOL_LIB; else export SHAPEMOL_LIB=$v; fi
  rm -rf gpurun_out/abp_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abp_$v -- python3 bench.py --batch ${BATCH:-256} --steps 100 --warmup 20 --cpu-steps 0 --concurrent 0 --exact-steps 0 --profile-steps 0 > gpurun_out/abp_$v.log 2>&1 || { echo "$v FAILED" >> gpurun_out/abp.txt; continue; }
  python - "$v" >> gpurun_out/abp.txt <<'PY'
import csv,glob,sys
v=sys.argv[1]
f=glob.glob(f"gpurun_out/abp_{v}/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
out=[]
for r in rows:
    n=r["Name"]
    for key in ("edge_stream_kernel<128, 8, false>","edge_stream_kernel<128, 8, true>","node_chain6","node_linear6","node_prologue6","graph_kernel","ddpm"):
        if key in n: out.append(f"{key.replace('edge_stream_kernel','es')}={float(r['AverageNs'])/1000:.2f}us x{r['Calls']}")
print(v, " ".join(out))
PY
  rm -rf gpurun_out/abp_$v
done
cat gpurun_out/abp.txt
