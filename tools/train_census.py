#!/usr/bin/env python3
"""From a rocprofv3 kernel trace (csv) of `bench.py --mode train`: kernel time per category and the top kernels, per step.
usage: train_census.py <rocprof dir> <steps executed in the run (timed + warm-up + capture warm-ups)>"""
import collections
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2])
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
cat_of = (("Cijk", "library GEMM"), ("k_gemm_f16x3", "own: split GEMM"), ("k_lin_f16x3", "own: layer GEMM"),
          ("k_cap_train", "own: captioner train"), ("k_lstm_train", "own: captioner train"),
          ("k_bwd_t1d", "own: msda bwd"), ("k_fwd_t1d", "own: msda fwd"), ("k_sum_partials", "own: msda bwd"),
          ("anonymous namespace)::k_", "own: other"), ("gvl::", "own: other"), ("multi_tensor_apply", "adam / clip"),
          ("layer_norm", "torch layer norm"), ("reduce_kernel", "torch reduce"), ("elementwise", "torch elementwise"),
          ("Cat", "torch cat/copy"), ("copy", "torch cat/copy"), ("fill", "torch fill"), ("Fill", "torch fill"),
          ("dropout", "torch dropout"), ("softmax", "torch softmax"), ("attn", "attention"), ("index", "torch index"))
tot, cnt, ktot, kcnt = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
for r in csv.DictReader(open(f)):
    name, us = r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    c = next((v for k, v in cat_of if k in name), "other")
    tot[c] += us
    cnt[c] += 1
    ktot[name] += us
    kcnt[name] += 1
print(f"# per step: {sum(tot.values()) / steps / 1e3:.2f} ms of kernel time in {sum(cnt.values()) / steps:.0f} launches")
for c, us in tot.most_common():
    print(f"{c:26s} {cnt[c] / steps:7.1f} launches {us / steps:8.1f} us")
print()
for name, us in ktot.most_common(45):
    print(f"{name[:120]:120s} {kcnt[name] / steps:7.1f} {us / steps:8.1f} us")
