#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of `bench.py --mode eval`: what the eval step spends OUTSIDE the captioner's token loop
(per step: kernel, launches, total us).  Usage: eval_rest_census.py <rocprof output dir> <steps in the run>"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
steps = sys.argv[2] if len(sys.argv) > 2 else "auto"
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
loop = ("k_gemm_f16x3", "k_vocab_f16x3", "k_gates_f16x3", "k_cap_attend", "k_lstm_cell", "k_greedy_from_partials", "k_greedy_and_gemm")
tot, cnt, loop_us = collections.Counter(), collections.Counter(), 0.0
for r in csv.DictReader(open(f)):
    name, us = r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if any(k in name for k in loop):
        loop_us += us
        continue
    tot[name] += us
    cnt[name] += 1
if steps == "auto":          # one k_pyramid_geometry launch per forward (inference layers); else the caller must say
    steps = max(1, sum(v for k, v in cnt.items() if "k_pyramid_geometry" in k))
steps = float(steps)
rest = sum(tot.values())
print(f"# forwards in the trace: {steps:.0f} (graph replays of the timed region + warm-up / capture / instrumented eager forwards)")
print(f"# per step: token loop {loop_us / steps / 1e3:.2f} ms, everything else {rest / steps / 1e3:.2f} ms in {sum(cnt.values()) / steps:.0f} launches")
for name, us in tot.most_common(60):
    print(f"{name[:110]:110s} {cnt[name] / steps:7.1f} launches {us / steps:8.1f} us")
