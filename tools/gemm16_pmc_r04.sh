#!/bin/bash
# PMC record of the vocabulary product -> gpurun_out/r04_gemm16_pmc.txt (one rocprofv3 --pmc pass per counter set, kernel trace only)
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/g16 && mkdir -p /tmp/g16
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/g16/pmc_$i -- python3 $root/tools/gemm16_pmc_r04.py > /dev/null 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/g16/trace -- python3 $root/tools/gemm16_pmc_r04.py > /dev/null 2>&1
python3 - <<'PY' > $root/gpurun_out/r04_gemm16_pmc.txt
import collections, csv, glob
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("/tmp/g16/pmc_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_gemm_f16x3_m16" not in k:
            continue
        form = "1 product (X1)" if "Lb1E" in k else "3 products"
        acc[form][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/g16/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_gemm_f16x3_m16" in r["Kernel_Name"]:
            dur["1 product (X1)" if "Lb1E" in r["Kernel_Name"] else "3 products"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# rocprofv3 --pmc (one pass per counter set, --kernel-trace only) over tools/gemm16_pmc_r04.py: the vocabulary product 4800 x 512 x 8518,")
print("# fused argmax form (k_gemm_f16x3_m16, 256 persistent workgroups x 8 wavefronts), stage-major operand planes; three launches per form, averaged")
for form in ("3 products", "1 product (X1)"):
    v = {n: sum(x) / len(x) for n, x in acc[form].items()}
    d = dur[form]
    print(f"== {form}: kernel {sum(d) / max(len(d), 1):.1f} us (kernel trace, back to back, min {min(d) if d else 0:.1f})")
    for n in sorted(v):
        print(f"   {n:32s} {v[n]:16.0f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_WAVE_CYCLES" in v:
        mf, wc = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024, v["SQ_WAVE_CYCLES"] * 4 / 2048
        print(f"   -> MFMA busy {mf / 1e3:.0f} K cycles per SIMD of {wc / 1e3:.0f} K kernel cycles per wavefront = {mf / wc:.2f}; "
              f"clock under load {wc / (sum(d) / len(d)) / 1e3:.2f} GHz")
    if "TCC_HIT_sum" in v:
        print(f"   -> L2 hit rate {v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.3f}; operand requests into L2 {v.get('TCP_TCC_READ_REQ_sum', 0) / 1e6:.1f} M")
    if "FETCH_SIZE" in v:
        print(f"   -> HBM 2 x FETCH_SIZE + WRITE_SIZE = {(2 * v['FETCH_SIZE'] + v.get('WRITE_SIZE', 0)) * 1024 / 1e6:.1f} MB")
PY
cat $root/gpurun_out/r04_gemm16_pmc.txt
