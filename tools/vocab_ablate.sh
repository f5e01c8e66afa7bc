#!/bin/bash
# dev experiment: what the fragment reads, the operand DMA, the stage barrier and the epilogue cost k_vocab_f16x3; timing-only builds
# (results wrong) of gvl_gemm16.hip alone, rebuilt and run alternately on one box (the GPU box's scratch copy: the shipped library
# is rebuilt at the end)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GVL_VOCAB_FORM=v
for d in "" "-DGVL_V_NO_EPI" "-DGVL_V_NO_EPI -DGVL_V_NO_DMA" "-DGVL_V_NO_EPI -DGVL_V_NO_VMWAIT" "-DGVL_V_NO_EPI -DGVL_V_SAME_SRC" \
         "-DGVL_V_NO_EPI -DGVL_V_SAME_SRC -DGVL_V_NO_VMWAIT" "-DGVL_V_NO_EPI -DGVL_V_NO_BARRIER" ""; do
  touch gvl_amd/csrc/gvl_gemm16.hip
  GVL_BUILD_DEFS="$d" python -c "from gvl_amd import build; build.build()" > /dev/null 2>&1
  echo "== defs '$d'"
  python tools/vocab_probe.py --time-only 2>&1 | grep "round [12]"
done
touch gvl_amd/csrc/gvl_gemm16.hip
python -c "from gvl_amd import build; build.build()" > /dev/null 2>&1
