#!/bin/bash
# A/B of the forward sampling kernel through an environment switch of the shipped library (e.g. GVL_MSDA_FWD_TRIP=0): per-workgroup
# phase stamps (in situ + back to back) and three interleaved eval bench lines each.  usage: tools/fwd_ab_env.sh NAME=VALUE
out=$PWD/gpurun_out/fwd_ab; mkdir -p $out
alt="$1"
python3 tools/fwd_phase_stamps.py 2>/dev/null | grep -v "^backward" > $out/stamps_shipped.txt
env $alt python3 tools/fwd_phase_stamps.py 2>/dev/null | grep -v "^backward" > $out/stamps_alt.txt
for k in 1 2 3; do
  env $alt python3 bench.py --mode eval --no-cpu-baseline --no-probes --steps 30 --warmup 5 2>/dev/null | tail -1 > $out/alt_$k.json
  python3 bench.py --mode eval --no-cpu-baseline --no-probes --steps 30 --warmup 5 2>/dev/null | tail -1 > $out/shipped_$k.json
done
echo "---- shipped"; cat $out/stamps_shipped.txt
echo "---- $alt"; cat $out/stamps_alt.txt
python3 - <<PY
import json, glob
for tag in ("shipped", "alt"):
    rows = []
    for f in sorted(glob.glob("$out/" + tag + "_*.json")):
        d = json.loads(open(f).read())
        r = d["roofline"]
        rows.append((d["value"], r.get("kernel_us"), r.get("frac"), (r.get("encoder_launch") or {}).get("kernel_us")))
    print(tag, rows)
PY
