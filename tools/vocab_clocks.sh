#!/bin/bash
# dev experiment: the shader clock k_vocab_f16x3 runs at, with and without its operand DMA (timing-only builds, GVL_V_CLOCKS prints
# workgroup 0's cycle and real-time counters)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GVL_VOCAB_FORM=v
for d in "-DGVL_V_CLOCKS" "-DGVL_V_CLOCKS -DGVL_V_DMA_ONCE" "-DGVL_V_CLOCKS -DGVL_V_NO_DMA" "-DGVL_V_CLOCKS -DGVL_V_SAME_SRC"; do
  touch gvl_amd/csrc/gvl_gemm16.hip
  GVL_BUILD_DEFS="$d" python -c "from gvl_amd import build; build.build()" > /dev/null 2>&1
  echo "== defs '$d'"
  python tools/vocab_probe.py --time-only --reps 30 2>&1 | grep "round 2\|k_vocab clocks" | tail -4
done
touch gvl_amd/csrc/gvl_gemm16.hip
python -c "from gvl_amd import build; build.build()" > /dev/null 2>&1
