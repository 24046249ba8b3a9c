#!/bin/bash
# usage (GPU box): tools/exp_rank.sh -- one rank's share of an N-GPU frame, k_frame against the per-level launches
for f in f1 f4; do
  echo "== k_frame $f"; timeout -k 10 200 python tools/rank_emulation.py $f 1 2 4 8 2>&1 | grep ranks
  echo "== per level $f"; T3D_PER_LEVEL=1 timeout -k 10 200 python tools/rank_emulation.py $f 1 2 4 8 2>&1 | grep ranks
done
