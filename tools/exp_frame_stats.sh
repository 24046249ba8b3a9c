#!/bin/bash
# usage (GPU box): tools/exp_frame_stats.sh  -- k_frame round statistics and timing, one frame at a time, plus a kernel trace
T3D_FRAME_STATS=1 timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --frames-in-flight 1 2>gpurun_out/exp_fs.err > gpurun_out/exp_fs.json || { tail -5 gpurun_out/exp_fs.err; exit 1; }
grep k_frame gpurun_out/exp_fs.err | tail -2
python - <<PY
import json
d=json.load(open("gpurun_out/exp_fs.json")); print("ms/frame", d["ms_per_step"], "Mrays/s", d["value"], "frame alone ms", d["pipeline"]["frame_latency_ms"], "frac", d["roofline"]["frac"])
PY
