#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_pmc_pass.sh <tag> <workload> <counter>...
# ONE counter pass (keep within the per-block limits) over a short bench run of one workload
tag=$1; wl=$2; shift 2
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcp_$tag
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmcp_$tag -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --frames-in-flight 1 > /dev/null 2> $R/gpurun_out/pmcp_$tag.err || { echo pass failed; tail -5 $R/gpurun_out/pmcp_$tag.err; exit 1; }
cd $R
python tools/pmc_summary.py gpurun_out/pmcp_$tag/*/*counter_collection.csv
