#!/bin/bash
# quick eval-step kernel table under rocprofv3 -> gpurun_out/peq.txt
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/peq
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/peq -- python3 $root/bench.py --mode eval --no-cpu-baseline --no-probes --steps 20 --warmup 5 > /dev/null 2>&1
cd $root
python tools/prof_summary.py /tmp/peq 14 > gpurun_out/peq.txt 2>&1
cut -c1-60,100-160 gpurun_out/peq.txt
