#!/bin/bash
# scratch: per-kernel average durations under library options (rocprofv3 --kernel-trace --stats, B = 256, 100 steps)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
: > gpurun_out/optp.txt
i=0
for o in "$@"; do
  i=$((i+1)); args=""; for kv in $o; do args="$args --opt $kv"; done
  rm -rf gpurun_out/optp_$i
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/optp_$i -- python3 bench.py --batch ${BATCH:-256} --steps 100 --warmup 20 --cpu-steps 0 --concurrent 0 --exact-steps 0 --profile-steps 0 $args > gpurun_out/optp_$i.log 2>&1 || { echo "$o FAILED" >> gpurun_out/optp.txt; continue; }
  python - "$o" $i >> gpurun_out/optp.txt <<'PY'
import csv,glob,sys
f=glob.glob(f"gpurun_out/optp_{sys.argv[2]}/**/*kernel_stats.csv",recursive=True)[0]
out=[]
for r in csv.DictReader(open(f)):
    n=r["Name"]
    for key in ("edge_stream_kernel<128, 8, false>","edge_stream_kernel<128, 8, true>","node_chain6","node_linear6","node_prologue6","graph_kernel","ddpm"):
        if key in n: out.append(f"{key.replace('edge_stream_kernel','es')}={float(r['AverageNs'])/1000:.2f}us x{r['Calls']}")
print(sys.argv[1], "|", " ".join(out))
PY
  rm -rf gpurun_out/optp_$i
done
cat gpurun_out/optp.txt
