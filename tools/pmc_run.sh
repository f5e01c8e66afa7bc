#!/bin/bash
# HBM traffic of the temporal kernels from PMC counters: separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE) per
# configuration, --kernel-trace only beside them (MI355X_MICROARCH.md, HBM / rocprofv3) -> gpurun_out/pmc/<cfg>_<counter>/
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for cfg in "100 f32" "100 f32 amax" "512 f32" "512 bf16"; do
  tag=$(echo $cfg | tr ' ' '_')
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${tag}_$ctr
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$ctr -- python3 $root/tools/pmc_target.py $cfg > /tmp/pmc_${tag}_$ctr.log 2>&1
  done
done
cd $root
python tools/pmc_traffic3.py /tmp > gpurun_out/r06_pmc_traffic.json
cat gpurun_out/r06_pmc_traffic.json | head -80
