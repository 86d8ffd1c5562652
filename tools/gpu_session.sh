cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_errors.jsonl
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 bash tools/profile_round.sh gpurun_out/prof_r04 > gpurun_out/prof_r04.log 2>&1; tail -2 gpurun_out/prof_r04.log
timeout -k 10 120 python tools/driver_bench.py > gpurun_out/driver_bench.json 2> gpurun_out/driver_bench.log
timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_tests.log; tail -4 gpurun_out/gpu_tests.log | cut -c1-200
