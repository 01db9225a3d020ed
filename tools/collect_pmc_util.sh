#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/collect_pmc_util.sh <tag> [workload]'): issue / matrix-pipe / LDS
# utilisation counters of every kernel of the training step (SURVEY.md 8d counter list), one rocprofv3 --pmc pass per
# counter group (the SQ block has 8 slots; --pmc is never combined with a trace domain other than --kernel-trace).
# The program goes directly after `--`.  Summary: gpurun_out/profiles/<tag>_pmc_util.json (tools/summarize_pmc_util.py);
# copy it into profiles/.
set -u
TAG=${1:-r02}
WL=${2:-c3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PMC_GROUPS=(
  "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE"
)
run_pass() {   # $1 = pass name, $2 = counters
  rocprofv3 --pmc $2 --kernel-trace --output-format csv -d "$OUT/util_$1" -o "$TAG" -- python3 "$ROOT/bench.py" \
    --no-cpu-baseline --no-secondary --no-public-path --workload $WL --no-graph --steps 20 --warmup 5 > /dev/null 2> "$OUT/util_$1.err"
  find "$OUT/util_$1" -name "${TAG}_counter_collection.csv" | grep -q .
}
i=0
for G in "${PMC_GROUPS[@]}"; do
  if ! run_pass "g$i" "$G"; then        # an unknown counter name fails the whole pass: fall back to one pass per counter
    echo "[pmc_util] group $i failed, retrying counter by counter" >&2
    tail -3 "$OUT/util_g$i.err" >&2
    for C in $G; do run_pass "g${i}_$C" "$C" || echo "[pmc_util] counter $C not collected" >&2; done
  fi
  i=$((i + 1))
done
python3 "$ROOT/tools/summarize_pmc_util.py" "$OUT" "$TAG" "$WL"
rm -rf "$OUT"/util_g*/ 2>/dev/null
ls -la "$OUT"
