#!/bin/bash
# dev experiment: what the operand DMA, the stage barrier and the epilogue cost k_vocab_f16x3; timing-only builds (results wrong) of
# gvl_gemm16.hip as tools/_bin/libgvl_msda_dev.so (the shipped library is not touched), run alternately on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GVL_VOCAB_FORM=v
for d in "" "-DGVL_V_NO_EPI" "-DGVL_V_NO_EPI -DGVL_V_NO_DMA" "-DGVL_V_NO_EPI -DGVL_V_NO_VMWAIT" "-DGVL_V_NO_EPI -DGVL_V_SAME_SRC" \
         "-DGVL_V_NO_EPI -DGVL_V_SAME_SRC -DGVL_V_NO_VMWAIT" "-DGVL_V_NO_EPI -DGVL_V_NO_BARRIER" ""; do
  python -m gvl_amd.build --dev gvl_gemm16.hip $d > /dev/null 2>&1
  echo "== defs '$d'"
  GVL_LIB_PATH=tools/_bin/libgvl_msda_dev.so python tools/vocab_probe.py --time-only 2>&1 | grep "round [12]"
done
