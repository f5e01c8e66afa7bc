#!/bin/bash
# PMC counters of the token step's two dominant launches (tools/gemm16_pmc_r06.py): one rocprofv3 --pmc pass per counter set,
# --kernel-trace only beside them (MI355X_MICROARCH.md, HBM / rocprofv3), plus one plain --kernel-trace pass for the durations
#   -> gpurun_out/r06p/gemm16_pmc.txt
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out/r06p
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/g16 && mkdir -p /tmp/g16
rocprofv3 --kernel-trace --output-format csv -d /tmp/g16/pmc_trace -- python3 $root/tools/gemm16_pmc_r06.py > /tmp/g16/trace.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/g16/pmc_$i -- python3 $root/tools/gemm16_pmc_r06.py > /tmp/g16/pmc_$i.log 2>&1
done
cd $root
python tools/gemm16_pmc_r06_summary.py /tmp/g16 > gpurun_out/r06p/gemm16_pmc.txt
cat gpurun_out/r06p/gemm16_pmc.txt
