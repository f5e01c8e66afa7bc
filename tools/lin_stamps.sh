#!/bin/bash
# dev experiment: where k_lin_f16x3's time goes (prologue / K loop / epilogue of one workgroup, s_memrealtime ticks); timing-only build
# as tools/_bin/libgvl_msda_dev.so
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m gvl_amd.build --dev gvl_layers.hip -DGVL_LIN_STAMPS > /dev/null 2>&1
GVL_LIB_PATH=tools/_bin/libgvl_msda_dev.so python tools/lin_bench.py 2>&1 | grep "k_lin" | sort | uniq -c | sort -rn | head -40
