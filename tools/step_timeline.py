#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of `bench.py --mode train|eval`: the ORDERED launches of one replayed step late in the run
(start offset, duration, gap to the previous launch's end), totals of busy / idle time and the same by kernel family.
Usage: step_timeline.py <rocprof output dir> <marker substring> [which step from the end, default 3] [--full]
The marker is a kernel launched exactly once per step (train: gvl_advance_step / k_advance_step; eval: k_pyramid_geometry)."""
import collections
import csv
import glob
import re
import sys

d, marker = sys.argv[1], sys.argv[2]
back = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 3
full = "--full" in sys.argv
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
if len(marks) < back + 2:
    sys.exit(f"marker {marker!r} seen {len(marks)} times")
# replayed (hipGraph) steps are the short ones; instrumented / eager / warm-up steps are several times longer on the wall:
# take the step of median wall time among those within 1.5x of the shortest (or, with "last", the back-th from the end)
if "--last" in sys.argv:
    lo, hi = marks[-back - 1], marks[-back]
else:
    iv = [(rows[b - 1][1] - rows[a][0], a, b) for a, b in zip(marks[:-1], marks[1:]) if b - a > 20]
    fast = sorted(x for x in iv if x[0] <= 1.5 * min(iv)[0])
    print(f"# {len(iv)} steps in the trace, {len(fast)} within 1.5x of the shortest ({fast[0][0] / 1e3:.0f} us); showing the median one")
    _, lo, hi = fast[len(fast) // 2]
step = rows[lo:hi]


def family(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    if n.startswith("Cijk_"):
        return "Tensile " + "_".join(n.split("_")[1:3]) + (" Bias" if "_Bias_" in n else "")
    if n.startswith("at::native::"):
        m = re.search(r"at::native::(?:\w+::)*(\w+)(?:<[^>]*?(\w+Functor|\w+Op|\w+_kernel|\w+Kernel)[^>]*)?", n)
        return "ATen " + (m.group(1) if m else n[:40]) + ((" " + m.group(2)) if m and m.group(2) else "")
    return re.split(r"[(<]", n)[0]


t0 = step[0][0]
busy = sum(e - s for s, e, _ in step)
wall = step[-1][1] - t0
print(f"# {f}\n# one step = launches [{lo}, {hi}) of {len(rows)}; {len(step)} launches, wall {wall / 1e3:.1f} us, "
      f"kernel-busy {busy / 1e3:.1f} us, idle {max(0, wall - busy) / 1e3:.1f} us")
fam_t, fam_n, fam_gap = collections.Counter(), collections.Counter(), collections.Counter()
prev_end = t0
lines = []
for s, e, n in step:
    gap = s - prev_end
    fam = family(n)
    fam_t[fam] += e - s
    fam_n[fam] += 1
    fam_gap[fam] += max(0, gap)
    lines.append(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.2f} {gap / 1e3:7.2f}  {re.sub(r'^void ', '', n)[:130]}")
    prev_end = max(prev_end, e)
print(f"{'family':70s} {'n':>5s} {'busy_us':>9s} {'gap_before_us':>13s}")
for fam, t in fam_t.most_common():
    print(f"{fam[:70]:70s} {fam_n[fam]:5d} {t / 1e3:9.1f} {fam_gap[fam] / 1e3:13.1f}")
if full:
    print("\n# start_us   dur_us  gap_us  kernel")
    print("\n".join(lines))
