#!/bin/bash
# Run ON THE GPU BOX: every rocprofv3 summary profiles/ holds for one round, workload by workload
# (gpurun -- 'bash tools/collect_all_profiles.sh r06').  Every step is bounded by `timeout`.
R=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for PAIR in final:c3 mimic:mimic c5:c5 c5m:c5m haim:haim c1:c1 c2:c2; do
  TAG=${R}_${PAIR%%:*}; WL=${PAIR##*:}
  timeout 600 bash "$ROOT/tools/collect_profiles.sh" "$TAG" "$WL" > "$ROOT/gpurun_out/collect_$TAG.log" 2>&1
  timeout 600 bash "$ROOT/tools/collect_pmc_util.sh" "$TAG" "$WL" >> "$ROOT/gpurun_out/collect_$TAG.log" 2>&1
done
timeout 120 python3 "$ROOT/tools/kernel_resources.py" > "$ROOT/gpurun_out/profiles/${R}_kernel_resources.txt" 2>/dev/null
ls "$ROOT/gpurun_out/profiles" | head -80
