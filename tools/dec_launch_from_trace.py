#!/usr/bin/env python3
"""Per-launch durations of the temporal deformable-attention kernels from a rocprofv3 --kernel-trace CSV of `bench.py`:
the four forward launches of a step come in the order encoder, encoder, decoder, decoder (the backward's in the reverse
order); reports each position's mean over the run, and over its second half (= the timed hipGraph replays).
Run bench.py with --no-probes for this (the supplementary B = 64 / cfg L kernel probes launch the same kernels).
usage: python tools/dec_launch_from_trace.py <output dir of rocprofv3 | ..._kernel_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

path = sys.argv[1]
if os.path.isdir(path):
    path = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = defaultdict(list)
with open(path) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        for key in ("k_fwd_t1d_d64", "k_bwd_t1d_split", "k_bwd_t1d_d64", "k_sum_partials"):
            if key in n:
                rows[key].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
                break
print(f"# source: {path}")
for key, order in (("k_fwd_t1d_d64", ("encoder", "encoder", "decoder", "decoder")),
                   ("k_bwd_t1d_split", ("decoder", "decoder", "encoder", "encoder")),
                   ("k_bwd_t1d_d64", ("decoder", "decoder", "encoder", "encoder")),
                   ("k_sum_partials", ("decoder", "decoder", "encoder", "encoder"))):
    rs = sorted(rows.get(key, []))
    if not rs:
        continue
    pos = defaultdict(list)
    for i, (a, b) in enumerate(rs):
        pos[i % 4].append((b - a) / 1e3)
    for k in range(4):
        v = pos[k]
        tail = v[len(v) // 2:]
        print(f"{key} launch {k} ({order[k]}): n={len(v)} mean {sum(v) / len(v):6.2f} us | second half of the run mean "
              f"{sum(tail) / len(tail):6.2f} us  min {min(v):6.2f}")
