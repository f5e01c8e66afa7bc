#!/bin/bash
# dev experiment: phases of k_gates_f16x3 (timing build with in-kernel stamps: prologue / K loop / cell / stores of one workgroup)
cd ${GRAFT_REPO_ROOT:-/root/repo}
python tools/gates_probe.py --reps 100 2>&1 | grep "round [12]"
touch gvl_amd/csrc/gvl_gemm16.hip
GVL_BUILD_DEFS="-DGVL_G_STAMPS" python -c "from gvl_amd import build; build.build()" > /dev/null 2>&1
python tools/gates_probe.py --reps 10 2>&1 | grep "k_gates" | tail -4
touch gvl_amd/csrc/gvl_gemm16.hip
python -c "from gvl_amd import build; build.build()" > /dev/null 2>&1
