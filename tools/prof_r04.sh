#!/bin/bash
# the round's profile set -> gpurun_out/r04p/ (copied into profiles/ afterwards)
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r04p; mkdir -p $out
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
python tools/bwd_own_probe.py > $out/bwd_long_video_own_vs_chunked.txt 2>&1
(tools/_bin/launch_floor; tools/_bin/valu_cost; tools/_bin/clock_check) > $out/ubench_launch_valu_clock.txt 2>&1
(python tools/msda_ab_probe.py fwd_amax - GVL_MSDA_XCD_PAIRS=0; python tools/msda_ab_probe.py bwd -; python tools/bwd_phase_stamps.py) > $out/msda_back_to_back.txt 2>&1
bash tools/pmc_run.sh > $out/pmc_run.log 2>&1
cp gpurun_out/r04_pmc_traffic.json $out/pmc_traffic.json
head -4 $out/eval_outside_token_loop.txt; tail -3 $out/msda_back_to_back.txt
