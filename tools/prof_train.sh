#!/bin/bash
# the training half of prof_round.sh alone (one short gpurun call) -> gpurun_out/r06p/: kernel stats, the ordered launches of one
# replayed step, the eager step's ATen census; `python tools/aten_by_kind.py gpurun_out/r06p/train_timeline.txt` summarises the tail
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r06p; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python3 $root/bench.py --mode train --no-cpu-baseline --no-probes --steps 20 --warmup 5 > $out/train_bench_line_under_rocprof.json 2> $out/train.err
cd $root
python tools/step_timeline.py /tmp/pt k_advance_step --full > $out/train_timeline.txt 2>&1
python tools/prof_summary.py /tmp/pt 45 > $out/train_kernel_stats.txt 2>&1
python tools/train_aten_sites.py > $out/train_aten_sites.txt 2>&1
head -3 $out/train_timeline.txt
