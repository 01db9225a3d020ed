# Diagnostic: per-call cost and per-step slope of MultiModN._train_steps for (first group, later groups) sizes
for cfg in "8 0" "8 12" "8 16" "8 24" "4 16"; do set -- $cfg; echo "== REPLAY_GROUP=$1 NEXT=$2"; MMN_RG=$1 MMN_RGN=$2 python tools/time_call_overhead.py 2>&1 | tail -7; done
