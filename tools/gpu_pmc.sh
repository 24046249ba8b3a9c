#!/bin/bash
# usage: tools/gpu_pmc.sh <tag>   -- two counter passes over a short bench run
tag=$1
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_${tag}_a -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_${tag}_a.err || { echo pass a failed; tail -5 $R/gpurun_out/pmc_${tag}_a.err; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM SQ_IFETCH GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${tag}_b -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/pmc_${tag}_b.err || { echo pass b failed; tail -5 $R/gpurun_out/pmc_${tag}_b.err; exit 1; }
cd $R
python tools/pmc_summary.py gpurun_out/pmc_${tag}_a/*/*counter_collection.csv gpurun_out/pmc_${tag}_b/*/*counter_collection.csv
