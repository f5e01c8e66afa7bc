#!/bin/bash
# rocprofv3 kernel trace of `bench.py --mode eval` -> gpurun_out/<tag>_rest.txt (what the step spends outside the token
# loop) and gpurun_out/<tag>_stats.txt (per-kernel totals).  usage: tools/prof_eval.sh <tag> [extra bench args]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $root/bench.py --mode eval --no-cpu-baseline --no-probes --steps 20 --warmup 5 "$@" > $root/gpurun_out/${tag}_bench.log 2>&1
cd $root
python tools/eval_rest_census.py /tmp/prof_$tag 25 > gpurun_out/${tag}_rest.txt 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -40 "$f" | cut -c1-220 > gpurun_out/${tag}_stats.txt
tail -1 gpurun_out/${tag}_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('under rocprof:', d['value'], d['ms_per_step'])"
