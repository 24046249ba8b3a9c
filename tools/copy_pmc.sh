#!/bin/bash
# usage (in the build container, after tools/gpu_pmc.sh <tag>, <tag>_cornell2_8k, <tag>_suzanne4k, <tag>_200k, <tag>_lbvh1m have run on the
# GPU box): tools/copy_pmc.sh <round> <tag>  -- copies the five counter summaries into profiles/, stamped with the commit they were made on
set -e
r=$1; tag=$2
c=$(git log --oneline | head -1 | cut -d' ' -f1)
for pair in "$tag:$r" "${tag}_cornell2_8k:${r}_cornell2_8k" "${tag}_suzanne4k:${r}_suzanne4k" "${tag}_200k:${r}_200k" "${tag}_lbvh1m:${r}_lbvh1m"; do
  s=${pair%%:*}; d=${pair##*:}
  python - "$s" "$d" "$c" <<'PY'
import json, shutil, sys
s, d, c = sys.argv[1:]
j = json.load(open(f"gpurun_out/pmc_{s}_traffic.json")); j["commit"] = c
json.dump(j, open(f"profiles/{d}_pmc_traffic.json", "w"), indent=1)
shutil.copy(f"gpurun_out/pmc_{s}_summary.txt", f"profiles/{d}_pmc_summary.txt")
PY
done
