#!/bin/bash
# One gpurun call's worth of validation on the MI355X box (the snapshot excludes gpurun_out/, so the session script lives here):
#   /usr/local/graft/bin/gpurun --timeout 1190 -- 'bash tools/gpu_session.sh'
# smoke, the full GPU suite (both precision modes), the round's profile set (tools/profile_round.sh) and the driver bench;
# everything lands under gpurun_out/ and is copied into profiles/<round>/ by hand afterwards.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_errors.jsonl
timeout -k 10 200 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 bash tools/profile_round.sh gpurun_out/prof_round > gpurun_out/prof_round.log 2>&1; tail -2 gpurun_out/prof_round.log
timeout -k 10 120 python tools/driver_bench.py > gpurun_out/driver_bench.json 2> gpurun_out/driver_bench.log
timeout -k 10 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/gpu_tests.log; tail -4 gpurun_out/gpu_tests.log | cut -c1-200
