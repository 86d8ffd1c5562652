#!/bin/bash
# Throughput regime on one GPU: configs[4] analogue (512 molecules of 40-80 atoms, k = 32) and B = 1024 (configs[2]/[3] per-GPU
# batch) with the two multi-job forms of the edge kernels (edge_tiles 1 = looping launch [default], 0 = sliced one-job launches)
#   tools/k32_sweep.sh > profiles/r03/throughput_sweep.txt
cd "$(dirname "$0")/.."
for t in 1 0; do
  echo "== k=32, B=512 x 40-80 atoms, edge_tiles=$t"
  python tools/kbench.py --batch 512 --atoms 40,80 --k 32 --steps 3 --graph-steps 20 --opt edge_tiles=$t 2>&1 | grep -E "edge_|node_|vn_|graph replay|atoms="
done
for t in 1 0; do
  echo "== k=8, B=1024 MOSES prior, edge_tiles=$t"
  python tools/kbench.py --batch 1024 --steps 3 --graph-steps 30 --opt edge_tiles=$t 2>&1 | grep -E "edge_|node_|graph replay"
done
