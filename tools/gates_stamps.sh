#!/bin/bash
# dev experiment: phases of k_gates_f16x3 (timing build with in-kernel stamps as tools/_bin/libgvl_msda_dev.so: prologue / K loop / cell /
# stores of one workgroup)
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/gates_probe.py --reps 100 2>&1 | grep "round [12]"
python -m gvl_amd.build --dev gvl_gemm16.hip -DGVL_G_STAMPS > /dev/null 2>&1
GVL_LIB_PATH=tools/_bin/libgvl_msda_dev.so python tools/gates_probe.py --reps 10 2>&1 | grep "k_gates" | tail -4
