#!/bin/bash
# train step with each round-6 switch turned off in turn, one box, one call:   bash tools/switch_sweep.sh [bench.py arguments]
# (which of the round's nodes pays at a configuration other than cfg A)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env "$@" python bench.py --mode train --no-cpu-baseline --no-probes --steps 16 --warmup 4 $ARGS 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('train_step_ms'))
except Exception as e: print('failed', e)"; }
ARGS="$*"
echo "all on: $(run X=1) $(run X=1)"
for s in GVL_WGRAD_SKIP=0 GVL_XT_MIRROR=0 GVL_CONV_SPLITK=0 GVL_LEVEL_POS=0 GVL_MASK_ROWS=0 GVL_RDLN_FAN=0 GVL_INPROJ_SHARED=0 GVL_EXPAND_PARTS=0 \
         GVL_COUNT_POOL=0 GVL_CLASS_COUNT_HEADS=0 GVL_CAP_SLAB=torch GVL_CAP_TIME_MAJOR=0 GVL_EMBED_ROWS=0 GVL_WGRAD_GROUP=0; do
  echo "$s: $(run $s)"
done
echo "all on: $(run X=1)"
