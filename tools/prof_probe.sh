#!/bin/bash
# run a probe script under rocprofv3 and print its kernel stats:  prof_probe.sh <script.py> [rows]
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $root/$1 > /tmp/pp.out 2>&1
tail -${3:-15} /tmp/pp.out
cd $root
python tools/prof_summary.py /tmp/pp ${2:-12} | cut -c1-60,100-150
