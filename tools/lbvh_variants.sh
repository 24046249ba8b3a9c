#!/bin/bash
# usage (on the GPU box, from the repo root): tools/lbvh_variants.sh <lib.so>...  -- the three LBVH workloads per library build
set -o pipefail
for lib in "$@"; do
  for w in suzanne_4k_b4 synthetic_200k_1080p_b2 synthetic_1m_4k_b2; do
    T3D_HIP_LIB=$PWD/$lib timeout -k 10 300 python -u bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2> /dev/null > gpurun_out/variant.json || exit 1
    python - <<PY
import json
d=json.load(open("gpurun_out/variant.json"))
r=d["roofline"]
print("$lib", "$w", "ms/frame", d["ms_per_step"], "nodes/query", r.get("nodes_visited_per_query"))
PY
  done
done
