#!/bin/bash
# eval forward with each inference switch of round 6 turned off in turn, one box, one call:  bash tools/switch_sweep_eval.sh [bench.py arguments]
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env "$@" python bench.py --mode eval --no-cpu-baseline --no-probes --steps 24 --warmup 4 $ARGS 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('ms_per_step'))
except Exception as e: print('failed', e)"; }
ARGS="$*"
echo "all on: $(run X=1) $(run X=1)"
for s in GVL_EVAL_OVERLAP=0 GVL_FIRST_LAYER_CACHE=0 GVL_CONV_SPLITK=0 GVL_GREEDY_MERGED=0 GVL_ATTEND_PRE=0 GVL_MSDA_XCD_PAIRS=0; do
  echo "$s: $(run $s)"
done
echo "all on: $(run X=1)"
