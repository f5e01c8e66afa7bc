#!/bin/bash
# dev experiment: what the stage barrier, the operand DMA and the epilogue arithmetic cost the vocabulary product (k_gemm_f16x3_m16);
# timing-only builds (results wrong), rebuilt and run alternately on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for d in "" "-DGVL_ABLATE_EPI" "-DGVL_ABLATE_EPI -DGVL_ABLATE_DMA" "-DGVL_ABLATE_EPI -DGVL_ABLATE_BARRIER" "-DGVL_ABLATE_EPI -DGVL_ABLATE_DMA -DGVL_ABLATE_BARRIER" "-DGVL_ABLATE_EPI -DGVL_ABLATE_LDS" "-DGVL_ABLATE_EPI -DGVL_ABLATE_LDS -DGVL_ABLATE_DMA -DGVL_ABLATE_BARRIER" ""; do
  python -m gvl_amd.build --dev gvl_gemm16.hip $d > /dev/null 2>&1
  echo "== defs '$d'"
  GVL_LIB_PATH=tools/_bin/libgvl_msda_dev.so python tools/x1_probe.py 2>&1 | grep "argmax form" | cut -c1-30
done
