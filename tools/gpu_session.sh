cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 300 --warmup 20 --cpu-steps 0 --concurrent 0 --exact-steps 0 > gpurun_out/b_stream.json 2> gpurun_out/b_stream.log; tail -2 gpurun_out/b_stream.log
done
timeout -k 10 200 python bench.py --batch 1024 --steps 100 --warmup 20 --cpu-steps 0 --concurrent 0 --exact-steps 0 > gpurun_out/b_stream_1024.json 2> gpurun_out/b_stream_1024.log; tail -2 gpurun_out/b_stream_1024.log
