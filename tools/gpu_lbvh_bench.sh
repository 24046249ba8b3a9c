#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_lbvh_bench.sh <tag>  -- LBVH tests, then the three LBVH workloads
set -o pipefail
tag=$1
timeout -k 10 500 python -u -m pytest tests/test_gpu_lbvh.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/tests_lbvh_$tag.log 2>&1
rc=$?
tail -3 gpurun_out/tests_lbvh_$tag.log
if [ $rc -ne 0 ]; then grep -E "^(FAILED|E  )" gpurun_out/tests_lbvh_$tag.log | head -20; exit $rc; fi
for w in suzanne_4k_b4 synthetic_200k_1080p_b2 synthetic_1m_4k_b2; do
  timeout -k 10 300 python -u bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline 2> /dev/null > gpurun_out/lbvh_${tag}_$w.json || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/lbvh_${tag}_$w.json"))
r=d["roofline"]
print("$w", "ms/frame", d["ms_per_step"], "Mrays/s", d["value"], "nodes/query", r.get("nodes_visited_per_query"), "GB/s", r["achieved"], "deep pushes", r.get("deep_stack_pushes"))
PY
done
