#!/bin/bash
# A/B of the temporal backward at the cfg A decoder / encoder shapes (tools/bwd_phase_stamps.py: per-workgroup phase stamps and
# dispatch times, back to back), three interleaved rounds.  Two modes:
#   tools/bwd_ab.sh env NAME=VALUE     the shipped library with and without an environment switch (e.g. GVL_MSDA_BWD_FLAT=0)
#   tools/bwd_ab.sh dev                the shipped library against a timing build in tools/_bin/libgvl_msda_dev.so
#                                      (made before the call with `python -m gvl_amd.build --dev gvl_msda.hip -D...`)
# Results under gpurun_out/bwd_ab/.
out=$PWD/gpurun_out/bwd_ab; mkdir -p $out
if [ "$1" = "env" ]; then alt="$2"; else alt="GVL_LIB_PATH=$PWD/tools/_bin/libgvl_msda_dev.so"; fi
for k in 1 2 3; do
  python3 tools/bwd_phase_stamps.py 2>/dev/null > $out/shipped_$k.txt
  env $alt python3 tools/bwd_phase_stamps.py 2>/dev/null > $out/alt_$k.txt
done
echo "---- shipped"; cat $out/shipped_*.txt | cut -c1-230
echo "---- $alt"; cat $out/alt_*.txt | cut -c1-230
