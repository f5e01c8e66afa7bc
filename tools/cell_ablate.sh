#!/bin/bash
# dev experiment: what the LSTM-cell epilogue of the gate product costs (k_gemm_f16x3<64, kLstm>): its operand loads, its arithmetic;
# timing-only builds (results wrong), rebuilt and run alternately on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for d in "" "-DGVL_ABLATE_CELL_MATH" "-DGVL_ABLATE_CELL_LOADS" "-DGVL_ABLATE_CELL_LOADS -DGVL_ABLATE_CELL_MATH" ""; do
  python -m gvl_amd.build --dev gvl_gemm16.hip $d > /dev/null 2>&1
  echo "== defs '$d'"
  GVL_LIB_PATH=tools/_bin/libgvl_msda_dev.so python tools/x1_probe.py 2>&1 | grep "gate product\|h product" | cut -c1-110
done
