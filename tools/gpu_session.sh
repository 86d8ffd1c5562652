cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1100 bash tools/profile_round.sh gpurun_out/prof_r04 > gpurun_out/prof_r04.log 2>&1; tail -3 gpurun_out/prof_r04.log
timeout -k 10 120 python tools/driver_bench.py > gpurun_out/driver_bench.json 2> gpurun_out/driver_bench.log; tail -2 gpurun_out/driver_bench.json
