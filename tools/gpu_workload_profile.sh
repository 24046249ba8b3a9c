#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_workload_profile.sh <tag> <workload>
# bench.py of one workload under rocprofv3 --kernel-trace --stats; prints the per-kernel summary
set -o pipefail
tag=$1; wl=$2
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_${tag}_bench.json 2> $R/gpurun_out/prof_$tag.err
rc=$?
cd $R
if [ $rc -ne 0 ]; then echo "BENCH FAILED rc=$rc"; tail -20 gpurun_out/prof_$tag.err; exit $rc; fi
python tools/prof_summary.py gpurun_out/prof_$tag/*/*_kernel_trace.csv
python tools/prof_summary.py gpurun_out/prof_$tag/*/*_kernel_trace.csv --frame 1 | sed -n '/frame launch/,$p'
