#!/bin/bash
# bench.py at shapes around the BASELINE configurations (robustness of the shape-dependent choices: plans, thresholds, kernel forms):
# eval ms and train ms per step, one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for a in "--batch 8" "--batch 32" "--T 64" "--T 200" "--T 300" "--queries 100" "--batch 4 --T 1024" "--cfg anet_c3d_ssvg" "--cfg anet_tsp_msvg_dvc" "--cfg yc2_tsn_dvc --T 300 --queries 100 --batch 8"; do
  python bench.py --no-cpu-baseline --no-probes --steps 12 --warmup 3 $a 2>/tmp/shape.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a', '| eval', d.get('ms_per_step'), 'ms | train', d.get('train_step_ms'), 'ms | fwd frac', (d.get('roofline') or {}).get('frac'))
except Exception as e: print('$a', 'FAILED', e); print(open('/tmp/shape.err').read()[-600:])"
done
