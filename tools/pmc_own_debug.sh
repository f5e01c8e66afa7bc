#!/bin/bash
# dev: HBM traffic of k_bwd_t1d_own at T = 512 fp32 with one phase switched off (GVL_MSDA_OWN_DEBUG: 1 = no phase A work,
# 2 = no phase B work, 0 = the kernel as shipped)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for dbg in 0 4 5 6; do
  export GVL_MSDA_OWN_DEBUG=$dbg
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcd_${dbg}_$ctr
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmcd_${dbg}_$ctr -- python3 $root/tools/pmc_target.py 512 f32 > /tmp/pmcd_${dbg}_$ctr.log 2>&1
  done
  python3 - <<PY
import csv, glob
out = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pmcd_${dbg}_%s/**/*counter_collection.csv" % ctr, recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if "k_bwd_t1d_own" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    # launches in order: enc x4, dec x4 -> take the last of each four
    vals = [float(r["Counter_Value"]) for r in rows]
    out[ctr] = (vals[3], vals[7]) if len(vals) >= 8 else vals
print("dbg=${dbg}", {k: [round(x * 1024 / 1e6, 1) for x in v] for k, v in out.items()}, "MB (raw, enc / dec); corrected enc = %.1f MB" % ((2 * out["FETCH_SIZE"][0] + out["WRITE_SIZE"][0]) * 1024 / 1e6))
PY
done
