#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/collect_profiles.sh <tag>'): collects the rocprofv3
# evidence bench.py's roofline block refers to and writes the summaries under gpurun_out/profiles/.
#   1. --kernel-trace --stats of the default bench command (hipGraph replay)
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of an eager run (per-dispatch counters)
# Copy gpurun_out/profiles/* into profiles/ afterwards (tools/summarize_profiles.py does the maths).
set -u
TAG=${1:-r01}
WL=${2:-c3}          # bench.py --workload
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o "$TAG" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary --no-public-path --workload $WL --steps 200 > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/kt.err"
cp "$OUT/kt/${TAG}_kernel_stats.csv" "$OUT/${TAG}_kernel_stats.csv" 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o "$TAG" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary --no-public-path --workload $WL --no-graph --steps 20 --warmup 5 > /dev/null 2> "$OUT/pmc_$C.err"
done
python3 "$ROOT/tools/summarize_profiles.py" "$OUT" "$TAG"
rm -rf "$OUT/kt" "$OUT"/pmc_FETCH_SIZE/*/ 2>/dev/null
ls -la "$OUT"
