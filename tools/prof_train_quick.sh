#!/bin/bash
# quick train-step kernel table under rocprofv3 -> gpurun_out/ptq.txt
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptq
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ptq -- python3 $root/bench.py --mode train --no-cpu-baseline --no-probes --steps 20 --warmup 5 > /dev/null 2>&1
cd $root
python tools/prof_summary.py /tmp/ptq 30 > gpurun_out/ptq.txt 2>&1
grep -E "skinny|k_zero_fill|k_cap_train|k_lstm" gpurun_out/ptq.txt
