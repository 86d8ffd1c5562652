cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for thr in 1e-5 1e-4 5e-4 4e-3; do
timeout -k 10 300 python tools/pinned_chain.py --mode exact --pins tools/probes/pins_wide_tmp.npz --thr $thr > gpurun_out/pinned_thr_$thr.log 2>&1; echo "thr $thr"; grep -E "^mode|step  950|step  990|end " gpurun_out/pinned_thr_$thr.log | cut -c1-200
done
