#!/bin/bash
# usage (GPU box): tools/exp_pmc_compare.sh -- instruction counters of k_frame against the per-level kernels (one pass each)
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
for mode in frame level; do
  rm -rf $R/gpurun_out/pmcx_$mode
  if [ $mode = level ]; then export T3D_PER_LEVEL=1; else unset T3D_PER_LEVEL; fi
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmcx_$mode -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --frames-in-flight 1 > /dev/null 2> $R/gpurun_out/pmcx_$mode.err || { echo pass $mode failed; tail -5 $R/gpurun_out/pmcx_$mode.err; exit 1; }
  echo "== $mode"
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcx_$mode/*/*counter_collection.csv
done
