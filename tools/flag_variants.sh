#!/bin/bash
# usage (on the GPU box, from the repo root): tools/flag_variants.sh <lib.so>...  -- the headline workload per library build: one frame alone
# and four in flight (bench.py's secondary timings off), twice each
for rep in 1 2; do
for lib in "$@"; do
  for f in 1 4; do
    T3D_HIP_LIB=$PWD/$lib timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-secondary --frames-in-flight $f 2> /dev/null > gpurun_out/variant.json || { echo "$lib failed"; continue; }
    python - <<PY
import json
d=json.load(open("gpurun_out/variant.json")); r=d["roofline"]
print("$lib", "in flight", $f, "ms/frame", d["ms_per_step"], "k_level launch us", round(r["avg_launch_ms"]*1e3,1), "frame alone ms", r["frame_device_ms"])
PY
  done
done
done
