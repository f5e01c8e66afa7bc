#!/bin/bash
# dev (round 4): which deformable-attention kernels run in the bf16 (autocast) train step at cfg A, and how long they take,
# next to the fp32 step: rocprofv3 kernel trace of `bench.py --mode train [--dtype bf16]`
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for dt in bf16; do
  rm -rf /tmp/pbt_$dt
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pbt_$dt -- python3 $root/bench.py --mode train --dtype $dt --no-cpu-baseline --no-probes --steps 10 --warmup 3 > /tmp/pbt_$dt.log 2>&1
  echo "== $dt"; tail -c 600 /tmp/pbt_$dt.log | head -c 300; echo
  python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pbt_$dt/**/*kernel_trace.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    n = r["Kernel_Name"]
    if "t1d" in n or "sum_partials" in n or "k_bwd_generic" in n or "k_fwd_generic" in n:
        acc[(n[:110], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("LDS_Block_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
seq = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in csv.DictReader(open(f[0])) if "bwd_t1d" in r["Kernel_Name"]]
seq.sort()
print("bwd launches in time order (us):", " ".join(f"{d:.0f}" for _, d in seq))
rows = [r for r in csv.DictReader(open(f[0]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tcut = next((int(r["Start_Timestamp"]) for r in rows if "bwd_t1d" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 31000), None)
print("first slow backward launch at", tcut)
if tcut:
    import collections as C
    before, after = C.defaultdict(list), C.defaultdict(list)
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        (before if int(r["Start_Timestamp"]) < tcut else after)[r["Kernel_Name"][:90]].append(d)
    tab = []
    for k in before:
        if k in after and len(before[k]) >= 20 and len(after[k]) >= 20:
            b, a_ = sorted(before[k]), sorted(after[k])
            tab.append((a_[len(a_) // 2] / b[len(b) // 2], b[len(b) // 2], a_[len(a_) // 2], len(b), len(a_), k))
    tab.sort(reverse=True)
    for t in tab[:14] + tab[-4:]:
        print(f"  x{t[0]:5.2f}  {t[1]:7.1f} -> {t[2]:7.1f} us  ({t[3]} / {t[4]} launches)  {t[5]}")
for k, v in sorted(acc.items()):
    v.sort()
    print(f"{len(v):5d} x  median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f}  max {v[-1]:7.2f}   {k}")
    if "bwd" in k[0]:
        import collections as C
        h = C.Counter(int(x // 2) * 2 for x in v)
        print("        2-us bins:", " ".join(f"{b}:{c}" for b, c in sorted(h.items())))
PY
done
