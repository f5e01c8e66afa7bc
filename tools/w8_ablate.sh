#!/bin/bash
# dev experiment: the [h2att ; W_hh] product (k_gemm_f16x3_w8<2,4,1,kStore>, 4800 x 512 x 2560) without its operand DMA / stage barrier /
# fragment reads; timing-only builds (results wrong), rebuilt and run alternately on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for d in "" "-DGVL_ABLATE_DMA" "-DGVL_ABLATE_BARRIER" "-DGVL_ABLATE_LDS" "-DGVL_ABLATE_LDS -DGVL_ABLATE_DMA -DGVL_ABLATE_BARRIER" ""; do
  GVL_BUILD_DEFS="$d" python -c "from gvl_amd import build; build.build(force=True)" > /dev/null 2>&1
  echo "== defs '$d'"
  python tools/x1_probe.py 2>&1 | grep "h product" | cut -c1-150
done
python -c "from gvl_amd import build; build.build(force=True)" > /dev/null 2>&1
