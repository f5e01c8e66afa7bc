#!/usr/bin/env python3
"""Dev tool: from a rocprofv3 kernel trace of `bench.py --mode train`, ONE steady-state step (between the last two Adam launches):
every __amd_rocclr_copyBuffer with the kernels around it and its grid size, aggregated.
usage: train_copy_context.py <rocprof dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
adam = [i for i, n in enumerate(names) if "multi_tensor_apply" in n and "Adam" in n or "FusedAdam" in n]
if len(adam) < 8:
    adam = [i for i, n in enumerate(names) if "multi_tensor_apply" in n]
# group consecutive optimizer launches; a step = from the end of one group to the end of the next
ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] - i > 50]
a, b = ends[-2] + 1, ends[-1] + 1
step = rows[a:b]
print("kernels in the step:", len(step), " copyBuffer:", sum("copyBuffer" in r["Kernel_Name"] for r in step),
      " wall us:", (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3)
short = lambda n: n.split("(")[0].replace("void ", "").replace("at::native::", "")[:70]      # noqa: E731
ctx = collections.Counter()
dur = collections.defaultdict(float)
for i, r in enumerate(step):
    if "copyBuffer" in r["Kernel_Name"]:
        prev = short(step[i - 1]["Kernel_Name"]) if i else "-"
        nxt = short(step[i + 1]["Kernel_Name"]) if i + 1 < len(step) else "-"
        g = r.get("Grid_Size_X") or r.get("Grid_Size") or "?"
        key = (prev, nxt, g)
        ctx[key] += 1
        dur[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for (prev, nxt, g), n in ctx.most_common(60):
    print(f"{n:3d} x grid {g:>8s} {dur[(prev, nxt, g)] / n:6.2f} us | after {prev:70s} | before {nxt}")

print("\n---- ATen kernels of the step by name + grid (count, total us)")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    n = r["Kernel_Name"]
    if "at::native" not in n and "rocclr" not in n:
        continue
    import re
    m = re.search(r"(\w+(Functor|_kernel_cuda|Op|_kernel|Kernel)\w*)", n.split("at::native::")[-1])
    key = (n.split("<")[0].replace("void at::native::", "")[:40], re.sub(r"std::array.*", "", n.split("at::native::")[-1])[:110],
           r.get("Grid_Size_X") or r.get("Grid_Size") or "?")
    agg[key][0] += 1
    agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for key, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{c:3d} x {t:7.1f} us  grid {key[2]:>9s}  {key[0]:40s} {key[1]}")
