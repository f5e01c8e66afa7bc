#!/bin/bash
# ordered per-step kernel timelines of the train and eval steps -> gpurun_out/r05t/
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r05t; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pe /tmp/pt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python3 $root/bench.py --mode train --no-cpu-baseline --no-probes --steps 20 --warmup 5 > $out/train_line.json 2> $out/train.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe -- python3 $root/bench.py --mode eval --no-cpu-baseline --no-probes --steps 20 --warmup 5 > $out/eval_line.json 2> $out/eval.err
cd $root
python tools/step_timeline.py /tmp/pt k_advance_step --full > $out/train_timeline.txt 2>&1
python tools/step_timeline.py /tmp/pe k_pyramid_geometry --full > $out/eval_timeline.txt 2>&1
python tools/prof_summary.py /tmp/pt 60 > $out/train_kernel_stats.txt 2>&1
head -3 $out/train_timeline.txt; head -3 $out/eval_timeline.txt
