cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 python tools/mode_check.py --batch 256 --oracle > gpurun_out/mc256.log 2>&1; tail -3 gpurun_out/mc256.log
timeout -k 10 200 python tools/mode_check.py --batch 1024 > gpurun_out/mc1024.log 2>&1; tail -1 gpurun_out/mc1024.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "exact and (ragged or variants or k32_b64 or other_k or extreme or taps or b4_golden or folded or chain_b4 or hash_noise)" > gpurun_out/t1.log 2>&1; tail -3 gpurun_out/t1.log
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 300 --warmup 20 --cpu-steps 0 --concurrent 0 --exact-steps 0 > gpurun_out/b_stream.json 2> gpurun_out/b_stream.log; tail -2 gpurun_out/b_stream.log
done
timeout -k 10 200 python bench.py --batch 1024 --steps 100 --warmup 20 --cpu-steps 0 --concurrent 0 --exact-steps 0 > gpurun_out/b_stream_1024.json 2> gpurun_out/b_stream_1024.log; tail -2 gpurun_out/b_stream_1024.log
timeout -k 10 200 python bench.py --batch 512 --atoms 40,80 --knn 32 --steps 40 --cpu-steps 0 --concurrent 0 --exact-steps 0 > gpurun_out/b_k32.json 2> gpurun_out/b_k32.log; tail -2 gpurun_out/b_k32.log
