#!/bin/bash
# like prof_msda.sh with free bench arguments: tools/prof_msda2.sh <tag> <bench args...>
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/pm_$tag -- python3 $root/bench.py --no-cpu-baseline --no-probes --steps 10 --warmup 3 "$@" > $root/gpurun_out/${tag}_bench.log 2>&1
cd $root
python tools/dec_launch_from_trace.py /tmp/pm_$tag > gpurun_out/${tag}_msda.txt 2>&1
cat gpurun_out/${tag}_msda.txt
