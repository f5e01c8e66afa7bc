#!/bin/bash
# Same-box comparison of two trees: the shipped one and a copy of an older revision under _r4tree/ (made HERE before the call:
#   git archive <rev> | tar -x -C _r4tree && (cd _r4tree && python -c "import __graft_entry__ as g; g.build()") ).
# Bench lines interleaved (eval + train, no CPU leg); results under gpurun_out/round_ab/.
out=$PWD/gpurun_out/round_ab; mkdir -p $out
for k in 1 2; do
  (cd _r4tree && python3 bench.py --no-cpu-baseline --steps 24 --warmup 3 2>/dev/null | tail -1) > $out/old_$k.json
  python3 bench.py --no-cpu-baseline --steps 24 --warmup 3 2>/dev/null | tail -1 > $out/new_$k.json
done
