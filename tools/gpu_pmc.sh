#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_pmc.sh <tag> [workload]
# four counter passes over a short bench run (each pass its own run; no trace domains mixed in)
tag=$1
wl=${2:-cornell_1080p_b4}
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
pass() {  # name, counters...
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_${tag}_$name -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --frames-in-flight 1 > /dev/null 2> $R/gpurun_out/pmc_${tag}_$name.err || { echo pass $name failed; tail -5 $R/gpurun_out/pmc_${tag}_$name.err; exit 1; }
}
pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY &&
pass b SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_IFETCH GRBM_GUI_ACTIVE &&
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE || exit 1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_${tag}_a/*/*counter_collection.csv gpurun_out/pmc_${tag}_b/*/*counter_collection.csv gpurun_out/pmc_${tag}_fetch/*/*counter_collection.csv gpurun_out/pmc_${tag}_write/*/*counter_collection.csv --workload $wl --traffic-json gpurun_out/pmc_${tag}_traffic.json | tee gpurun_out/pmc_${tag}_summary.txt
