#!/usr/bin/env python3
"""Summarise tools/msda_pmc_r06.sh: per launch shape (forward / backward x encoder / decoder at cfg A) the mean of every counter over the
three measured launches, per wavefront where that is the natural unit, and the issue shares they imply."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/pmc_[0-9]*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "t1d" in r["Kernel_Name"]]
    # dispatch order: per kernel family 4 launches per shape (the first a warm-up), shapes enc then dec
    by_counter = collections.defaultdict(list)
    for r in rows:
        by_counter[r["Counter_Name"]].append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    for c, lst in by_counter.items():
        lst.sort()
        fam = collections.defaultdict(list)
        for _, k, v in lst:
            fam["fwd" if "k_fwd" in k else "bwd"].append(v)
        for kind, vals in fam.items():
            for si, shape in enumerate(("encoder (Lq = 188)", "decoder (Lq = 300)")):
                part = vals[4 * si + 1:4 * si + 4]
                if part:
                    acc[(kind, shape)][c] = part
print("# rocprofv3 --pmc, one pass per counter set (--kernel-trace only beside it), over tools/pmc_target.py 100 f32 amax; the mean of three launches")
print("# per shape.  SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count in units of 4 cycles per wavefront (the r06_gemm16_pmc.txt convention).")
for key in (("fwd", "encoder (Lq = 188)"), ("fwd", "decoder (Lq = 300)"), ("bwd", "encoder (Lq = 188)"), ("bwd", "decoder (Lq = 300)")):
    c = acc.get(key)
    if not c:
        continue
    v = {n: sum(x) / len(x) for n, x in c.items()}
    print(f"== {key[0]} {key[1]}")
    for n in sorted(v):
        print(f"   {n:28s} {v[n]:16.0f}")
    w = v.get("SQ_WAVES")
    if w and "SQ_INSTS_VALU" in v:
        print(f"   -> per wavefront: {v['SQ_INSTS_VALU'] / w:.0f} vector-ALU instructions, {v.get('SQ_INSTS_LDS', 0) / w:.0f} LDS, "
              f"{v.get('SQ_INSTS_VMEM_RD', 0) / w:.0f} vector loads, {v.get('SQ_INSTS_VMEM_WR', 0) / w:.0f} vector stores, {v.get('SQ_INSTS_SALU', 0) / w:.0f} scalar")
    if "SQ_WAVE_CYCLES" in v and "SQ_ACTIVE_INST_VALU" in v:
        wc = v["SQ_WAVE_CYCLES"]
        print(f"   -> of a wavefront's resident cycles: issuing vector ALU {v['SQ_ACTIVE_INST_VALU'] / wc:.2f}, LDS {v.get('SQ_ACTIVE_INST_LDS', 0) / wc:.2f}, "
              f"vector memory {v.get('SQ_ACTIVE_INST_VMEM', 0) / wc:.2f}; any instruction {v.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f}, "
              f"issue-stalled {v.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} (on LDS {v.get('SQ_WAIT_INST_LDS', 0) / wc:.2f}), parked at a wait / barrier {v.get('SQ_WAIT_ANY', 0) / wc:.2f}")
    if "SQ_LDS_BANK_CONFLICT" in v and v.get("SQ_ACTIVE_INST_LDS"):
        print(f"   -> LDS bank-conflict cycles / LDS active cycles: {v['SQ_LDS_BANK_CONFLICT'] / v['SQ_ACTIVE_INST_LDS']:.2f}")
