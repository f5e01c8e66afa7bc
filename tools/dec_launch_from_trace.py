#!/usr/bin/env python3
"""Per-launch durations of k_fwd_t1d_d64 from a rocprofv3 --kernel-trace CSV of `bench.py` (eval): the four launches of
a forward come in the order encoder, encoder, decoder, decoder; reports each position's mean over the run.
usage: python tools/dec_launch_from_trace.py <..._kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if "k_fwd_t1d_d64" in r["Kernel_Name"] and "true, true, false" in r["Kernel_Name"].replace("1, 1, 0", "true, true, false"):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
pos = defaultdict(list)
for i, (a, b) in enumerate(rows):
    pos[i % 4].append((b - a) / 1e3)
for k in range(4):
    v = pos[k]
    tail = v[len(v) // 2:]                    # second half of the run = timed hipGraph replays
    print(f"launch {k} ({'encoder' if k < 2 else 'decoder'}): n={len(v)} mean {sum(v) / len(v):6.2f} us | "
          f"second half of the run mean {sum(tail) / len(tail):6.2f} us  min {min(v):6.2f}")
