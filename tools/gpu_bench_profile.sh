#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_bench_profile.sh <tag>
# runs the GPU tests, then bench.py under rocprofv3 --kernel-trace --stats, prints a summary
set -o pipefail
tag=$1
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/tests_$tag.log 2>&1
rc=$?
tail -3 gpurun_out/tests_$tag.log
if [ $rc -ne 0 ]; then echo "TESTS FAILED rc=$rc"; grep -E "^(FAILED|E  )" gpurun_out/tests_$tag.log | head -20; exit $rc; fi
# default command: two frames in flight (kernels of the two chains overlap, so their durations include sharing the device)
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_${tag}_bench.json 2> $R/gpurun_out/prof_$tag.err
rc=$?
cd $R
if [ $rc -ne 0 ]; then echo "BENCH FAILED rc=$rc"; tail -20 gpurun_out/prof_$tag.err; exit $rc; fi
echo "== default command (frames in flight: see the bench line) =="
python tools/prof_summary.py gpurun_out/prof_$tag/*/*_kernel_trace.csv | sed -n '1,/^$/p'
# one frame at a time and nothing but whole frames (--no-secondary: no half-frame bands, no other seed mode): every kernel alone
# on the device, the launches the roofline figures are about
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${tag}_solo -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --frames-in-flight 1 > $R/gpurun_out/prof_${tag}_solo_bench.json 2> $R/gpurun_out/prof_${tag}_solo.err
rc=$?
cd $R
if [ $rc -ne 0 ]; then echo "BENCH FAILED rc=$rc"; tail -20 gpurun_out/prof_${tag}_solo.err; exit $rc; fi
echo "== --frames-in-flight 1 =="
python tools/prof_summary.py gpurun_out/prof_${tag}_solo/*/*_kernel_trace.csv
python tools/prof_summary.py gpurun_out/prof_${tag}_solo/*/*_kernel_trace.csv --frame 1 | sed -n '/frame launch/,$p' | sed 's/frame launch sequence/untimed frame (no events between the launches), launch sequence/'
python - <<PY
import json
for name in ("prof_${tag}_bench.json", "prof_${tag}_solo_bench.json"):
    d=json.load(open("gpurun_out/" + name))
    print(name, "value", d["value"], "Mrays/s  ms/frame", d["ms_per_step"], " k_level frac", d["roofline"]["frac"], "avg launch ms", d["roofline"]["avg_launch_ms"], "pipeline", d.get("pipeline"))
PY
