#!/bin/bash
# configs[4] analogue on one GPU (512 molecules of 40-80 atoms, k = 32): occupancy sweep of the two-tile edge kernel
# (waves per workgroup; two 16-column tiles per wave-job, both MLP images resident = one workgroup per CU)
#   tools/k32_sweep.sh > profiles/r02_final/k32_sweep.txt
cd "$(dirname "$0")/.."
for w in 4 6 8; do
  echo "== k=32, B=512 x 40-80 atoms, edge_waves=$w"
  python tools/kbench.py --batch 512 --atoms 40,80 --k 32 --steps 3 --graph-steps 20 --opt edge_waves=$w 2>&1 | grep -E "edge_|node_|graph replay|atoms="
done
echo "== k=8, B=1024 MOSES prior: the three multi-job forms (edge_tiles 0 = sliced one-job launches, 1 = looping one-tile, 2 = looping two-tile)"
for t in 0 1 2; do
  echo "-- edge_tiles=$t"
  python tools/kbench.py --batch 1024 --steps 3 --graph-steps 30 --opt edge_tiles=$t 2>&1 | grep -E "edge_|graph replay"
done
