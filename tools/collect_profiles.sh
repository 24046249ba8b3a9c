#!/bin/bash
# usage (in the build container, after gpu_bench_profile.sh <tag>, gpu_round_numbers.sh <tag> and gpu_pmc.sh <pmctag> [+ _lbvh1m]
# have been run on the GPU box): tools/collect_profiles.sh <round> <tag> <pmctag>  -- copies the summaries into profiles/
set -e
r=$1; tag=$2; pmc=$3
G=gpurun_out; P=profiles
commit=$(git log --oneline | head -1 | cut -d' ' -f1)
sed -n '/^== default command/,$p' $G/${tag}_profile.log | grep -v "^prof_" > $P/${r}_bench_kernel_trace_summary.txt
cp "$(ls -t $G/prof_${tag}/*/*_kernel_stats.csv | head -1)" $P/${r}_bench_kernel_stats.csv   # (the newest run of that tag)
cp "$(ls -t $G/prof_${tag}_solo/*/*_kernel_stats.csv | head -1)" $P/${r}_bench_solo_kernel_stats.csv
cp $G/prof_${tag}_bench.json $P/${r}_bench_under_rocprof.json
cp $G/prof_${tag}_solo_bench.json $P/${r}_bench_solo_under_rocprof.json
cp $G/${tag}_bench_default.json $P/${r}_bench_default.json
for w in suzanne_4k_b4 synthetic_200k_1080p_b2 synthetic_1m_4k_b2 cornell2_8k_b8; do cp $G/${tag}_bench_$w.json $P/${r}_bench_$w.json; done
cp $G/${tag}_lbvh1m_kernel_trace_summary.txt $P/${r}_lbvh1m_kernel_trace_summary.txt
cp $G/${tag}_suzanne4k_kernel_trace_summary.txt $P/${r}_suzanne4k_kernel_trace_summary.txt
cp $G/prof_${tag}_lbvh1m_bench.json $P/${r}_lbvh1m_bench_under_rocprof.json
cp $G/${tag}_time_api.txt $P/${r}_time_api.txt
grep -E "^(ranks|   k_level)" $G/${tag}_rank_emulation.txt > $P/${r}_rank_emulation.txt
# counter runs: the headline workload under the plain names, the others with a suffix
for x in "" _lbvh1m _suzanne4k _200k _cornell2_8k; do
  [ -f "$G/pmc_${pmc}${x}_traffic.json" ] || continue
  python - "$G/pmc_${pmc}${x}_traffic.json" "$P/${r}${x}_pmc_traffic.json" "$commit" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); d["commit"] = sys.argv[3]
json.dump(d, open(sys.argv[2], "w"), indent=1)
PY
  cp $G/pmc_${pmc}${x}_summary.txt $P/${r}${x}_pmc_summary.txt
done
ls -la $P | grep " ${r}_"
