#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes over tools/gemm16_pmc.py (one directory per pass) per kernel form."""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_f16x3" not in k:
            continue
        form = ("argmax 4800x512x8518" if "ELi2EEE" in k.split("k_gemm")[1][:40] and "w8" in k else
                "store w8 " + r["Grid_Size"] if "w8" in k else "store 4-wave " + r["Grid_Size"])
        acc[form][r["Counter_Name"]].append(float(r["Counter_Value"]))
for form, c in acc.items():
    v = {n: sum(x) / len(x) for n, x in c.items()}
    print(f"== {form}")
    for n in sorted(v):
        print(f"   {n:32s} {v[n]:16.0f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CYCLES" in v:
        print(f"   MFMA busy per SIMD / kernel busy cycles (4 SIMDs x CUs aggregated as the counters report them): see DESIGN 4.17")
    if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
        print(f"   L2 hit rate {v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.3f}")
