#!/bin/bash
# dev experiment: the [h2att ; W_hh] product (k_gemm_f16x3_w8<2,4,1,kStore>, 4800 x 512 x 2560) without its operand DMA / stage barrier /
# fragment reads; timing-only builds (results wrong), rebuilt and run alternately on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for d in "" "-DGVL_ABLATE_DMA" "-DGVL_ABLATE_BARRIER" "-DGVL_ABLATE_LDS" "-DGVL_ABLATE_LDS -DGVL_ABLATE_DMA -DGVL_ABLATE_BARRIER" ""; do
  python -m gvl_amd.build --dev gvl_gemm16.hip $d > /dev/null 2>&1
  echo "== defs '$d'"
  GVL_LIB_PATH=tools/_bin/libgvl_msda_dev.so python tools/x1_probe.py 2>&1 | grep "h product" | cut -c1-150
done
