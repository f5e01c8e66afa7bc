#!/bin/bash
# A/B of the token loop with the offsets' hidden-state product inside the attention kernel (GVL_ATTEND_PRE=0: round 4's form) and
# riding in the h2att(h) launch (default): bench lines, interleaved, + per-kernel averages of the token loop from a kernel trace.
# Run on the GPU box from the repo root; results under gpurun_out/attend_pre/.
out=$PWD/gpurun_out/attend_pre; mkdir -p $out
for k in 1 2; do
  GVL_ATTEND_PRE=0 python3 bench.py --mode eval --no-cpu-baseline --no-probes --steps 30 --warmup 5 2>/dev/null | tail -1 > $out/off_$k.json
  python3 bench.py --mode eval --no-cpu-baseline --no-probes --steps 30 --warmup 5 2>/dev/null | tail -1 > $out/on_$k.json
done
cd /tmp && export TMPDIR=/tmp
for v in off on; do
  if [ $v = off ]; then export GVL_ATTEND_PRE=0; else unset GVL_ATTEND_PRE; fi
  rm -rf /tmp/ap_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ap_$v -- python3 /root/repo/bench.py --mode eval --no-cpu-baseline --no-probes --steps 10 --warmup 3 > /dev/null 2>&1
  f=$(find /tmp/ap_$v -name '*kernel_stats.csv' | head -1)
  grep -E "k_cap_attend|k_gates|k_vocab|k_greedy|k_gemm_f16x3" $f | awk -F, '{print $1, $2, $4}' | cut -c1-200 > $out/kernels_$v.txt
done
