#!/bin/bash
# dev experiment: what the LSTM-cell epilogue of the gate product costs (k_gemm_f16x3<64, kLstm>): its operand loads, its arithmetic;
# timing-only builds (results wrong), rebuilt and run alternately on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for d in "" "-DGVL_ABLATE_CELL_MATH" "-DGVL_ABLATE_CELL_LOADS" "-DGVL_ABLATE_CELL_LOADS -DGVL_ABLATE_CELL_MATH" ""; do
  GVL_BUILD_DEFS="$d" python -c "from gvl_amd import build; build.build(force=True)" > /dev/null 2>&1
  echo "== defs '$d'"
  python tools/x1_probe.py 2>&1 | grep "gate product\|h product" | cut -c1-110
done
python -c "from gvl_amd import build; build.build(force=True)" > /dev/null 2>&1
