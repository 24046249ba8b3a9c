#!/bin/bash
# usage (GPU box): tools/exp_frame_unit.sh  -- k_frame with different unit sizes against the per-level path, one frame at a time
for u in 128 64 32; do
  T3D_FRAME_STATS=1 T3D_FRAME_UNIT=$u timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --frames-in-flight 1 2>gpurun_out/exp_unit_$u.err > gpurun_out/exp_unit_$u.json || exit 1
  grep k_frame gpurun_out/exp_unit_$u.err | tail -1
  python - <<PY
import json
d=json.load(open("gpurun_out/exp_unit_$u.json")); print("unit $u: ms/frame", d["ms_per_step"], "Mrays/s", d["value"], "frame alone ms", d["pipeline"]["frame_latency_ms"])
PY
done
T3D_PER_LEVEL=1 timeout -k 10 120 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --frames-in-flight 1 2>/dev/null > gpurun_out/exp_unit_pl.json || exit 1
python - <<PY
import json
d=json.load(open("gpurun_out/exp_unit_pl.json")); print("per level: ms/frame", d["ms_per_step"], "Mrays/s", d["value"], "frame alone ms", d["pipeline"]["frame_latency_ms"])
PY
