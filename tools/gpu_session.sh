cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 python tools/mode_check.py --batch 256 --oracle > gpurun_out/mc256.log 2>&1; tail -3 gpurun_out/mc256.log
timeout -k 10 600 python tools/fuzz_parity.py --cases 40 > gpurun_out/fuzz_parity.txt 2>&1; tail -4 gpurun_out/fuzz_parity.txt
timeout -k 10 300 python tools/soak.py > gpurun_out/soak.txt 2>&1; tail -5 gpurun_out/soak.txt
