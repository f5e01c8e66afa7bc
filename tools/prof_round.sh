#!/bin/bash
# the round's profile set -> gpurun_out/r06p/ (copied into profiles/ afterwards): kernel stats + ordered per-step timelines of
# the train and eval steps from rocprofv3 kernel traces of `bench.py`, the MSDA launches inside the replayed graphs
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r06p; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pe /tmp/pt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python3 $root/bench.py --mode train --no-cpu-baseline --no-probes --steps 20 --warmup 5 > $out/train_bench_line_under_rocprof.json 2> $out/train.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -- python3 $root/bench.py --mode eval --no-cpu-baseline --no-probes --steps 20 --warmup 5 > $out/eval_bench_line_under_rocprof.json 2> $out/eval.err
cd $root
python tools/step_timeline.py /tmp/pt k_advance_step --full > $out/train_timeline.txt 2>&1
python tools/step_timeline.py /tmp/pe k_pyramid_geometry --full > $out/eval_timeline.txt 2>&1
python tools/prof_summary.py /tmp/pt 45 > $out/train_kernel_stats.txt 2>&1
python tools/prof_summary.py /tmp/pe 40 > $out/eval_kernel_stats.txt 2>&1
python tools/eval_rest_census.py /tmp/pe auto > $out/eval_outside_token_loop.txt 2>&1
python tools/dec_launch_from_trace.py /tmp/pe > $out/msda_launches_in_graph.txt 2>&1
python tools/dec_launch_from_trace.py /tmp/pt >> $out/msda_launches_in_graph.txt 2>&1
head -3 $out/train_timeline.txt; head -3 $out/eval_timeline.txt
