#!/bin/bash
# dev experiment: the eval step's kernels with the two-launch gate product of round 4 (GVL_GATES_FUSED=0) and with
# gvl_gemm_f16x3_gates_f32, from rocprofv3 kernel traces of `bench.py --mode eval` on one box -> gpurun_out/gates_ab/
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/gates_ab; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for g in ${1:-0 1}; do
  rm -rf /tmp/pe$g
  export GVL_GATES_FUSED=$g
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe$g -- python3 $root/bench.py --mode eval --no-cpu-baseline --no-probes --steps 20 --warmup 5 > $out/bench_$g.json 2> $out/err_$g.txt
  (cd $root; python tools/prof_summary.py /tmp/pe$g 8 > $out/stats_$g.txt 2>&1)
  head -12 $out/stats_$g.txt | cut -c1-200
done
