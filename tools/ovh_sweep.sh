for cfg in "8 0" "2 8" "1 8" "4 8"; do set -- $cfg; echo "== REPLAY_GROUP=$1 NEXT=$2"; MMN_RG=$1 MMN_RGN=$2 python tools/time_call_overhead.py 2>&1 | tail -6; done
