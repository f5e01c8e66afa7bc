#!/bin/bash
# counters of the captioner's attention kernel (plain | coarse levels in LDS): one rocprofv3 --pmc pass per counter set,
# --kernel-trace only beside them -> gpurun_out/r03_cap_attend_pmc.txt
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rm -rf /tmp/pmccap_$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmccap_$i -- python3 $root/tools/pmc_cap_target.py > /tmp/pmccap_$i.log 2>&1
done
cd $root
python - <<'PY' > gpurun_out/r03_cap_attend_pmc.txt
import collections, csv, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("/tmp/pmccap_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_cap_attend" not in k:
            continue
        acc["k_cap_attend_lds" if "_lds" in k else "k_cap_attend"][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --pmc, one pass per counter set (tools/pmc_cap.sh over tools/pmc_cap_target.py): 16 videos x 300 queries, T = 100;")
print("# mean over the 4 launches of each kernel.  FETCH_SIZE / WRITE_SIZE in KiB (HBM side, FETCH_SIZE x 2 per MI355X_MICROARCH.md).")
for kern, c in acc.items():
    print("==", kern)
    v = {n: sum(x) / len(x) for n, x in c.items()}
    for n in sorted(v):
        print(f"   {n:28s} {v[n]:16.0f}")
    if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
        print(f"   L2 hit rate {v['TCC_HIT_sum'] / max(1.0, v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.3f}")
PY
cat gpurun_out/r03_cap_attend_pmc.txt
