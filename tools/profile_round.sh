#!/bin/bash
# Every profile of a round in one go (on the MI355X box): kernel trace + stats of the graph-replay bench, the PMC passes
# (HBM traffic: FETCH_SIZE, WRITE_SIZE; three SQ counter sets) of a short eager run, the same at B = 1024, plain bench lines.
#   tools/profile_round.sh [outdir]      (default gpurun_out/prof_round)
# Counter passes run alone with --kernel-trace (never with other trace domains); rocprofv3 gets `python3 bench.py` directly.
set -eo pipefail
OUT=${1:-gpurun_out/prof_round}
mkdir -p "$OUT"
export TMPDIR=/tmp
SHORT="--steps 20 --warmup 5 --clock-warmup 0 --cpu-steps 0 --profile-steps 2 --eager --concurrent 0 --exact-steps 0"
find_csv() { find "$1" -name "*$2" | head -1; }
LIBSHA=$(sha1sum shapemol_amd/libshapemol_hip.so | cut -c1-16)

for B in 256 1024; do
  tag=$([ $B = 256 ] && echo "" || echo "_b$B")
  steps=$([ $B = 256 ] && echo 300 || echo 100)
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace$tag" -- python3 bench.py --batch $B --steps $steps --warmup 20 --cpu-steps 0 --concurrent 0 --exact-steps 0 \
      > "$OUT/bench_line${tag}_profiled.json" 2> "$OUT/trace$tag.err"
  cp "$(find_csv "$OUT/trace$tag" _kernel_stats.csv)" "$OUT/kernel_stats$tag.csv"
  [ $B = 256 ] && python3 tools/trace_gaps.py "$(find_csv "$OUT/trace$tag" _kernel_trace.csv)" > "$OUT/per_step_breakdown.txt"
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_${C}$tag" -- python3 bench.py --batch $B $SHORT > /dev/null 2> "$OUT/pmc_${C}$tag.err"
  done
  python3 tools/pmc_traffic.py "$(find_csv "$OUT/pmc_FETCH_SIZE$tag" _counter_collection.csv)" "$(find_csv "$OUT/pmc_WRITE_SIZE$tag" _counter_collection.csv)" "$LIBSHA" > "$OUT/pmc_traffic$tag.json"
  echo "B=$B done" >&2
done

{
  echo "# rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py $SHORT (B=256): mean per launch, summed over SEs/XCDs."
  echo "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x4 = cycles); SQ_VALU_MFMA_BUSY_CYCLES counts cycles."
} > "$OUT/sq_counters.txt"
i=0
for SET in "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_LDS"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/sq$i" -- python3 bench.py $SHORT > /dev/null 2> "$OUT/sq$i.err"
  python3 tools/pmc_summary.py "$(find_csv "$OUT/sq$i" _counter_collection.csv)" >> "$OUT/sq_counters.txt"
  echo >> "$OUT/sq_counters.txt"
done
echo "sq done" >&2

mkdir -p profiles/r04 && cp "$OUT"/pmc_traffic*.json profiles/r04/      # bench.py reads its `traffic` field from the committed summaries (keyed to the library's SHA-1)
python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_line_20steps.json" 2> "$OUT/bench_line_20steps.err"
python3 bench.py --batch 1024 --steps 100 --cpu-steps 0 > "$OUT/bench_line_b1024.json" 2> "$OUT/bench_line_b1024.err"
python3 bench.py --batch 512 --atoms 40,80 --knn 32 --steps 60 --cpu-steps 0 --concurrent 0 > "$OUT/bench_line_k32_b512.json" 2> "$OUT/bench_line_k32_b512.err"
# keep only the summaries (the raw traces are large)
rm -rf "$OUT"/trace* "$OUT"/pmc_FETCH* "$OUT"/pmc_WRITE* "$OUT"/sq[0-9]
ls -la "$OUT" >&2
