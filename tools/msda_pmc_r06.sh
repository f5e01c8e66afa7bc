#!/bin/bash
# issue / LDS counters of the sampling forward and backward at cfg A (tools/pmc_target.py 100 f32 amax: per shape 1 warm-up + 3 measured
# launches in the order fwd enc, fwd dec, bwd enc, bwd dec): one rocprofv3 --pmc pass per counter set, --kernel-trace only beside it
#   -> gpurun_out/r06p/msda_pmc.txt
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out/r06p
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/mp && mkdir -p /tmp/mp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/mp/pmc_$i -- python3 $root/tools/pmc_target.py 100 f32 amax > /tmp/mp/pmc_$i.log 2>&1
done
cd $root
python tools/msda_pmc_r06_summary.py /tmp/mp > gpurun_out/r06p/msda_pmc.txt
cat gpurun_out/r06p/msda_pmc.txt
