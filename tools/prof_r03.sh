#!/bin/bash
# the round's profile set -> gpurun_out/r03p/ (copied into profiles/ afterwards)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r03p; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pe /tmp/pt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -- python3 $root/bench.py --mode eval --no-cpu-baseline --no-probes --steps 20 --warmup 5 > $out/eval_bench_line_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python3 $root/bench.py --mode train --no-cpu-baseline --no-probes --steps 20 --warmup 5 > $out/train_bench_line_under_rocprof.json 2> /dev/null
cd $root
python tools/eval_rest_census.py /tmp/pe auto > $out/eval_outside_token_loop.txt 2>&1
python tools/prof_summary.py /tmp/pe 40 > $out/eval_kernel_stats.txt 2>&1
python tools/prof_summary.py /tmp/pt 45 > $out/train_kernel_stats.txt 2>&1
python tools/dec_launch_from_trace.py /tmp/pe > $out/msda_launches_in_graph.txt 2>&1
python tools/dec_launch_from_trace.py /tmp/pt >> $out/msda_launches_in_graph.txt 2>&1
(python tools/lin_bench.py; python tools/lin_bench.py --mha; python tools/lin_ksweep.py) > $out/layer_kernels.txt 2>&1
python tools/bwd_t512_probe.py > $out/bwd_t512_fp32_vs_bf16.txt 2>&1
head -4 $out/eval_outside_token_loop.txt
