#!/bin/bash
# dev experiment: where k_lin_f16x3's time goes (prologue / K loop / epilogue of one workgroup, s_memrealtime ticks); timing-only build
cd ${GRAFT_REPO_ROOT:-/root/repo}
touch gvl_amd/csrc/gvl_layers.hip
GVL_BUILD_DEFS="-DGVL_LIN_STAMPS" python -c "from gvl_amd import build; build.build()" > /dev/null 2>&1
python tools/lin_bench.py 2>&1 | grep "k_lin" | sort | uniq -c | sort -rn | head -40
touch gvl_amd/csrc/gvl_layers.hip
python -c "from gvl_amd import build; build.build()" > /dev/null 2>&1
