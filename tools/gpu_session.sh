cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_errors.jsonl
timeout -k 10 1100 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 > gpurun_out/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_tests.log; tail -25 gpurun_out/gpu_tests.log | cut -c1-220
