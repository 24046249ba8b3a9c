#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_quick_bench.sh [F ...]  -- bench.py with F frames in flight (default: 1 2)
fs=${@:-1 2}
for f in $fs; do
timeout -k 10 200 python -u bench.py --no-cpu-baseline --frames-in-flight $f 2> /dev/null | python -c '
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print("F", d["config"]["frames_in_flight"], "ms/frame", d["ms_per_step"], "Mrays/s", d["value"], "k_level avg launch ms", r["avg_launch_ms"], "frac", r["frac"], "frame alone ms", r["frame_device_ms"])' || exit 1
done
