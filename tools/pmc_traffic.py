#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; each over `python tools/opbench.py --iters 20`) into
profiles/rNN_pmc_traffic.json: HBM bytes per launch of the temporal kernels, corrected as MI355X_MICROARCH.md
(section HBM / rocprofv3) prescribes for gfx950 (FETCH_SIZE counts in KiB and reports half of the bytes of wide coalesced
reads: traffic = 2 * FETCH_SIZE + WRITE_SIZE).
    python tools/pmc_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> > profiles/r01_pmc_traffic.json"""
import csv
import glob
import json
import re
import sys


variants = set()


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    out = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"k_(fwd|bwd)_t1d_d64<(\d), (true|false), (true|false), (true|false), (?:(true|false), )?(\w+)>",
                      r["Kernel_Name"])
        if m:
            kind, pad, full, fused, l0g, _loop, vt = m.groups()
        else:
            # the level-split backward (two workgroups per slab, round 2): k_bwd_t1d_split<PAD, FUSED, VT>; it takes the
            # role (and the key) of k_bwd_t1d_d64 at the shapes it serves
            m = re.search(r"k_bwd_t1d_split<(\d), (true|false), (\w+)>", r["Kernel_Name"])
            if not m:
                continue
            kind, l0g = "bwd", "false"
            pad, fused, vt = m.groups()
            variants.add("k_bwd_t1d_split")
        if vt != "float" or l0g == "true":
            continue
        key = f"k_{kind}_t1d_d64" + ("_fused" if fused == "true" else "")
        out.setdefault(key, []).append(float(r["Counter_Value"]) * 1024.0)      # KiB -> bytes
    return out


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
res = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes (with --kernel-trace only) over "
               "`python tools/opbench.py --iters 20` (B=16,T=100,M=8,D=64,L=4,P=4); per kernel the max over dispatches is "
               "the decoder launch Lq=300, the min the encoder launch Lq=188; gfx950 correction per "
               "MI355X_MICROARCH.md: counters are in KiB and FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads "
               "-> traffic = 2*FETCH_SIZE + WRITE_SIZE (an upper estimate for the 4/8-byte operand streams). "
               "Summarised by tools/pmc_traffic.py."}
for key in sorted(fetch):
    for tag, pick in (("dec", max), ("enc", min)):
        f_, w_ = pick(fetch[key]), pick(write.get(key, [0.0]))
        res[f"{key}_{tag}"] = {"fetch_bytes_raw": f_, "write_bytes": w_, "hbm_bytes_corrected": 2 * f_ + w_}
import hashlib
import os
_src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gvl_amd", "csrc", "gvl_msda.hip")
res["kernel_source_sha16"] = hashlib.sha256(open(_src, "rb").read()).hexdigest()[:16]    # bench.py flags a stale file
res["backward_kernel_variant"] = sorted(variants) or ["k_bwd_t1d_d64 (+ k_sum_partials, not counted here)"]
res["algorithmic_bytes"] = {"fwd_dec": 23363584, "fwd_enc": 16941056, "bwd_dec": 36900000, "bwd_enc": 27720000}
print(json.dumps(res, indent=1))
