#!/bin/bash
# dev experiment: cost of the fused argmax epilogue's arithmetic in the vocabulary product (timing-only build, results wrong)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for d in "" "-DGVL_ABLATE_EPI" "" "-DGVL_ABLATE_EPI"; do
  python -m gvl_amd.build --dev gvl_gemm16.hip $d > /dev/null 2>&1
  echo "== defs '$d'"
  GVL_LIB_PATH=tools/_bin/libgvl_msda_dev.so python tools/x1_probe.py 2>&1 | grep "argmax form"
done
