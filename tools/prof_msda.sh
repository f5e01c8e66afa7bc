#!/bin/bash
# rocprofv3 kernel trace of `bench.py --mode <mode>` -> per-position durations of the deformable-attention launches
# usage: tools/prof_msda.sh <tag> <eval|train> [env assignments...]
tag=$1; mode=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/pm_$tag -- python3 $root/bench.py --mode $mode --no-cpu-baseline --no-probes --steps 20 --warmup 5 > $root/gpurun_out/${tag}_bench.log 2>&1
cd $root
python tools/dec_launch_from_trace.py /tmp/pm_$tag > gpurun_out/${tag}_msda.txt 2>&1
cat gpurun_out/${tag}_msda.txt
