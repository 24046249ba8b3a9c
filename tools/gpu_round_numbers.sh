#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_round_numbers.sh <tag>
# the numbers a round's write-up quotes, without a profiler attached: the default bench line, the three LBVH workloads,
# the API latencies, one rank's share of a several-GPU frame; then the LBVH workloads under rocprofv3 --kernel-trace
set -o pipefail
tag=$1
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -u bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err || { tail -5 gpurun_out/${tag}_bench_default.err; exit 1; }
cat gpurun_out/${tag}_bench_default.json
for w in suzanne_4k_b4 synthetic_200k_1080p_b2 synthetic_1m_4k_b2 cornell2_8k_b8; do
  timeout -k 10 400 python -u bench.py --workload $w --steps 3 --warmup 1 > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err || { tail -5 gpurun_out/${tag}_bench_$w.err; exit 1; }
  cat gpurun_out/${tag}_bench_$w.json
done
timeout -k 10 200 python -u tools/time_api.py > gpurun_out/${tag}_time_api.txt 2>&1 || { tail -5 gpurun_out/${tag}_time_api.txt; exit 1; }
cat gpurun_out/${tag}_time_api.txt
# one frame per call, then as bench.py runs several ranks: as many frames per call as there are ranks
timeout -k 10 200 python -u tools/rank_emulation.py f4 g 1 2 4 8 > gpurun_out/${tag}_rank_emulation.txt 2>&1 || { tail -5 gpurun_out/${tag}_rank_emulation.txt; exit 1; }
timeout -k 10 200 python -u tools/rank_emulation.py f4 bw g 2 4 8 >> gpurun_out/${tag}_rank_emulation.txt 2>&1 || { tail -5 gpurun_out/${tag}_rank_emulation.txt; exit 1; }
for w in suzanne_4k_b4 synthetic_1m_4k_b2; do
  timeout -k 10 300 python -u tools/rank_emulation.py $w f1 1 8 >> gpurun_out/${tag}_rank_emulation.txt 2>&1 || { tail -5 gpurun_out/${tag}_rank_emulation.txt; exit 1; }
  timeout -k 10 300 python -u tools/rank_emulation.py $w f1 bw 2 4 8 >> gpurun_out/${tag}_rank_emulation.txt 2>&1 || { tail -5 gpurun_out/${tag}_rank_emulation.txt; exit 1; }
done
cat gpurun_out/${tag}_rank_emulation.txt
bash tools/gpu_workload_profile.sh ${tag}_lbvh1m synthetic_1m_4k_b2 > gpurun_out/${tag}_lbvh1m_kernel_trace_summary.txt || exit 1
bash tools/gpu_workload_profile.sh ${tag}_suzanne4k suzanne_4k_b4 > gpurun_out/${tag}_suzanne4k_kernel_trace_summary.txt || exit 1
tail -30 gpurun_out/${tag}_lbvh1m_kernel_trace_summary.txt
