#!/usr/bin/env python3
"""Summarise tools/gemm16_pmc_r06.sh: per kernel the mean duration (plain kernel trace) and the mean of every counter, and what
they imply -- MFMA-busy ratio, the shader clock under load (GRBM_GUI_ACTIVE cycles / duration), bytes at the memory side of L2
(FETCH_SIZE doubled: MI355X_MICROARCH.md, HBM), L2 hit rate."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
names = {"k_vocab_f16x3": "k_vocab_f16x3  (4800 x 512 x 8518, argmax / sum-exp epilogue)",
         "k_gates_f16x3": "k_gates_f16x3  (n = 4800, 4H = 2048, K = 1024, LSTM cell epilogue)"}


def which(k):
    for n in names:
        if n in k:
            return n
    return None


dur = collections.defaultdict(list)
for f in glob.glob(root + "/pmc_trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = which(r["Kernel_Name"])
        if n:
            dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/pmc_[0-9]*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        n = which(r["Kernel_Name"])
        if n:
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --pmc (one pass per counter set, --kernel-trace only beside it) over tools/gemm16_pmc_r06.py; five launches of each kernel per")
print("# pass, the first dropped (cold), the rest averaged.  Durations from a plain --kernel-trace pass of the same program.")
wgs = {"k_vocab_f16x3": 256, "k_gates_f16x3": 240}       # persistent grids at these shapes (510 tiles over 256 CUs; 240 tiles)
flops = {"k_vocab_f16x3": 3 * 2 * 4800 * 512 * 8518, "k_gates_f16x3": 3 * 2 * 4800 * 1024 * 2048}
for n, title in names.items():
    d = sorted(dur[n])[: max(1, len(dur[n]) - 1)] if dur[n] else [float("nan")]
    us = sum(d) / len(d)
    print(f"== {title}: {us:.1f} us per launch (min {min(d):.1f}) = {flops[n] / us / 1e9:.3f} PFLOP/s of fp16 products "
          f"= {flops[n] / us / 1e9 / 2.5:.2f} of 2.5")
    v = {c: sum(x[1:]) / max(1, len(x) - 1) if len(x) > 1 else x[0] for c, x in acc[n].items()}
    for c in sorted(v):
        print(f"   {c:32s} {v[c]:16.0f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_WAVE_CYCLES" in v:
        # SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over the SIMDs (= MOPS / 2: a 16 x 16 x 32 fp16 MFMA is 32 MOPS and 16 cycles);
        # SQ_WAVE_CYCLES: a wavefront's resident time in units of 4 cycles, summed over the wavefronts (persistent: 8 per workgroup,
        # one workgroup per CU, alive from dispatch to the kernel's end)
        simds, waves = wgs[n] * 4, wgs[n] * 8
        busy, life = v["SQ_VALU_MFMA_BUSY_CYCLES"] / simds, 4 * v["SQ_WAVE_CYCLES"] / waves
        print(f"   -> MFMA pipe busy {busy / 1e3:.1f} K cycles per SIMD of {life / 1e3:.1f} K cycles a wavefront is resident = {busy / life:.2f}")
        print(f"   -> shader clock under load >= {life / us / 1e3:.2f} GHz (resident cycles / launch duration; 2.4 GHz nominal): the MFMA cycles "
              f"alone are {busy / (life / us) :.1f} us at this clock, {busy / 2.4e3:.1f} us at 2.4 GHz")
    if "GRBM_GUI_ACTIVE" in v:
        print(f"   -> GRBM_GUI_ACTIVE / 8 XCDs / duration = {v['GRBM_GUI_ACTIVE'] / us / 8e3:.2f} GHz (an upper bound: the counter also runs around the dispatch)")
    if "SQ_INSTS_VALU_MFMA_MOPS_F16" in v:
        print(f"   -> MFMA ops counted (F16 MOPS, 512 flops each): {v['SQ_INSTS_VALU_MFMA_MOPS_F16'] * 512 / 1e9:.1f} GFLOP against {flops[n] / 1e9:.1f} algorithmic")
    if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
        print(f"   -> L2 hit rate {v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.3f}")
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        print(f"   -> memory side of L2 (KB as reported; FETCH_SIZE doubled as the guide prescribes for 16-byte streaming reads on gfx950; Infinity-Cache "
              f"hits are counted too): {2 * v['FETCH_SIZE'] / 1e3:.1f} MB read, {v['WRITE_SIZE'] / 1e3:.1f} MB written")
