#!/bin/bash
# dev experiment: the shader clock k_vocab_f16x3 runs at, with and without its operand DMA (timing-only builds as
# tools/_bin/libgvl_msda_dev.so; GVL_V_CLOCKS prints workgroup 0's cycle and real-time counters)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GVL_VOCAB_FORM=v
for d in "-DGVL_V_CLOCKS" "-DGVL_V_CLOCKS -DGVL_V_DMA_ONCE" "-DGVL_V_CLOCKS -DGVL_V_NO_DMA" "-DGVL_V_CLOCKS -DGVL_V_SAME_SRC"; do
  python -m gvl_amd.build --dev gvl_gemm16.hip $d > /dev/null 2>&1
  echo "== defs '$d'"
  GVL_LIB_PATH=tools/_bin/libgvl_msda_dev.so python tools/vocab_probe.py --time-only --reps 30 2>&1 | grep "round 2\|k_vocab clocks" | tail -4
done
