cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 bash tools/profile_round.sh gpurun_out/prof_r04a > gpurun_out/prof_r04a.log 2>&1; tail -5 gpurun_out/prof_r04a.log
cat gpurun_out/prof_r04a/bench_line.json | head -c 3000
